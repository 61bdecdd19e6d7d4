#!/bin/bash
# everything the round's profiles/ entries are made from, in one GPU call: tools/final_round.sh <tag>
tag=${1:-r01_f}
cd $GRAFT_REPO_ROOT
o=gpurun_out/final_$tag
mkdir -p $o
bash tools/round_profile.sh $tag > $o/round_profile.txt 2>&1
timeout -k 10 300 python tools/op_times.py 2>&1 | grep -v amdgpu.ids > $o/op_times.txt
timeout -k 10 500 python tools/ks_time.py 256 --cpu 2>&1 | grep -v amdgpu.ids > $o/ks_time.txt
timeout -k 10 300 python tools/matmul_time.py 2>&1 | grep -v amdgpu.ids > $o/matmul_time.txt
timeout -k 10 300 python tools/encode_time.py --cpu 2>&1 | grep -v amdgpu.ids > $o/encode_time.txt
timeout -k 10 200 ./tools/cpp/bench_shim > $o/bench_shim.txt 2>&1
timeout -k 10 100 ./tools/valu_bench > $o/valu_bench.txt 2>&1
timeout -k 10 600 ./tools/cpp/bench_attention 16 768 > $o/attention.txt 2>&1
timeout -k 10 600 ./tools/cpp/bench_bootstrap_lt 32 16 > $o/bootstrap_lt.txt 2>&1
timeout -k 10 600 ./tools/cpp/bench_bootstrap 16 16 16 > $o/bootstrap_3.txt 2>&1
timeout -k 10 900 ./tools/cpp/bench_layer 16 > $o/layer_full.txt 2>&1
grep -a "LayerNorm\|intermediate\|GELU\|final X\|feed-forward" $o/layer_full.txt | cut -c1-400 > $o/layer.txt
timeout -k 10 300 ./tools/cpp/bench_streams 15 > $o/streams.txt 2>&1
tail -3 $o/round_profile.txt | cut -c1-200
cat $o/op_times.txt $o/ks_time.txt $o/matmul_time.txt $o/encode_time.txt $o/bench_shim.txt $o/attention.txt $o/bootstrap_lt.txt $o/bootstrap_3.txt $o/layer.txt $o/streams.txt
