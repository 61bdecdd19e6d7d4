#!/bin/bash
# End-of-round evidence on the GPU box: the bench line, the rocprofv3 kernel summary of the same command,
# and HBM traffic of the NTT kernels from separate --pmc passes.  Outputs under gpurun_out/round/.
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/round
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py > $out/${tag}_bench_line.json 2> $out/bench.err || echo "bench failed"
tail -1 $out/${tag}_bench_line.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-keyswitch > $out/stats.log 2>&1 || echo "stats failed"
cp $out/stats/*/*kernel_stats.csv $out/${tag}_bench_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-e2e --no-keyswitch --steps 2 --warmup 1 > $out/pmc_$c.log 2>&1 || echo "pmc $c failed"
done
mkdir -p $out/pmc_all && for c in FETCH_SIZE WRITE_SIZE; do mkdir -p $out/pmc_all/p_$c && cp -r $out/pmc_$c/* $out/pmc_all/p_$c/; done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc_all > $out/${tag}_pmc_summary.json
cat $out/${tag}_pmc_summary.json | head -40
head -8 $out/${tag}_bench_kernel_stats.csv | cut -c1-150
