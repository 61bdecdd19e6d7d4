cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -1 gpurun_out/pytest_gpu.log
bash tools/round_profile.sh r01_j > gpurun_out/round_profile_j.txt 2>&1
tail -12 gpurun_out/round_profile_j.txt | cut -c1-200
