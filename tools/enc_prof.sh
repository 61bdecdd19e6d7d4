set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/encprof; mkdir -p gpurun_out/encprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/encprof/r -- python3 tools/encode_time.py --big > gpurun_out/encprof/log.txt 2>&1
cat gpurun_out/encprof/log.txt | grep -v amdgpu
f=$(find gpurun_out/encprof/r -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-140
