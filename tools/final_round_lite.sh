#!/bin/bash
# the parts of tools/final_round.sh that the last changes of a round touch, sized for one 20-minute GPU call
tag=${1:-r01_i}
cd $GRAFT_REPO_ROOT
o=gpurun_out/final_$tag
mkdir -p $o
timeout -k 10 300 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; tail -1 $o/pytest_gpu.txt
bash tools/round_profile.sh $tag > $o/round_profile.txt 2>&1
timeout -k 10 300 python tools/op_times.py 2>&1 | grep -v amdgpu.ids > $o/op_times.txt
timeout -k 10 300 python tools/ks_time.py 256 2>&1 | grep -v amdgpu.ids > $o/ks_time.txt
timeout -k 10 600 ./tools/cpp/bench_attention 16 768 > $o/attention.txt 2>&1
timeout -k 10 600 ./tools/cpp/bench_bootstrap 16 16 16 > $o/bootstrap_3.txt 2>&1
tail -3 $o/round_profile.txt | cut -c1-200
cat $o/op_times.txt $o/ks_time.txt $o/attention.txt $o/bootstrap_3.txt
