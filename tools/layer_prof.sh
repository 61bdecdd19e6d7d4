set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/layerprof
rm -rf gpurun_out/layerprof/*
timeout -k 10 1000 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/layerprof -- tools/cpp/bench_encoder_layer > /dev/null 2> gpurun_out/layerprof/run.log || { tail -20 gpurun_out/layerprof/run.log; exit 1; }
grep -v "rocprofv3\|simple_timer\|output_stream" gpurun_out/layerprof/run.log | tail -20
f=$(find /tmp/layerprof -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -c1-170 > gpurun_out/layerprof/kernel_stats_head.csv
cat gpurun_out/layerprof/kernel_stats_head.csv
