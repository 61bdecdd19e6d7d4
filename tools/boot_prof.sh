set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bootprof
rm -rf gpurun_out/bootprof/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bootprof/p -- tools/cpp/bench_bootstrap ${1:-16} 16 0 > gpurun_out/bootprof/run.log 2>&1
cat gpurun_out/bootprof/run.log
f=$(find gpurun_out/bootprof/p -name "*kernel_stats.csv" | head -1)
head -30 $f | cut -c1-160 > gpurun_out/bootprof/kernel_stats_head.csv
cat gpurun_out/bootprof/kernel_stats_head.csv
find gpurun_out/bootprof/p -name "*kernel_trace.csv" -delete
