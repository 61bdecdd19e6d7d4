#!/bin/bash
# everything the round's profiles/ entries are made from, in one GPU call
cd $GRAFT_REPO_ROOT
bash tools/round_profile.sh r01_d > gpurun_out/round_profile_d.txt 2>&1
timeout -k 10 300 python tools/op_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/op_times_d.txt
timeout -k 10 500 python tools/ks_time.py 64 --cpu 2>&1 | grep -v amdgpu.ids > gpurun_out/ks_time_d.txt
timeout -k 10 300 python tools/matmul_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/matmul_time_d.txt
timeout -k 10 200 ./tools/cpp/bench_shim > gpurun_out/bench_shim_d.txt 2>&1
timeout -k 10 100 ./tools/valu_bench > gpurun_out/valu_bench_d.txt 2>&1
tail -3 gpurun_out/round_profile_d.txt | cut -c1-200; cat gpurun_out/op_times_d.txt gpurun_out/ks_time_d.txt gpurun_out/matmul_time_d.txt gpurun_out/bench_shim_d.txt
