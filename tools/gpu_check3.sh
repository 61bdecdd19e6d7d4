set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ct_pt_dot or ct_dot" > gpurun_out/dot_test.log 2>&1 || { tail -40 gpurun_out/dot_test.log; exit 1; }
tail -2 gpurun_out/dot_test.log
timeout -k 10 300 tests/cpp/test_bootstrap_lt > gpurun_out/lt.log 2>&1 || { tail -30 gpurun_out/lt.log; exit 1; }
cat gpurun_out/lt.log
