set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
tail -1 gpurun_out/pytest_gpu.log
MOAI_NTT_FP=0 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
timeout -k 10 400 tools/cpp/bench_bootstrap 48 16 16 2>&1 | grep "packed\|per bootstrap\|per ciphertext\|results"
python tools/ks_time.py --only 35 1 2>&1 | grep "L="
python tools/ks_time.py --only 35 64 2>&1 | grep "L="
