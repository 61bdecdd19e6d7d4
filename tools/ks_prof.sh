set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ksprof
rm -rf gpurun_out/ksprof/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksprof/fp -- python3 tools/ks_time.py --only 35 64 > gpurun_out/ksprof/fp.log 2>&1
MOAI_NTT_FP=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksprof/int -- python3 tools/ks_time.py --only 35 64 > gpurun_out/ksprof/int.log 2>&1
for v in fp int; do f=$(find gpurun_out/ksprof/$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -12 $f | cut -c1-120; done
