set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 300 tests/cpp/test_bootstrap_lt > gpurun_out/test_bootstrap_lt.log 2>&1 || { tail -30 gpurun_out/test_bootstrap_lt.log; exit 1; }
tail -1 gpurun_out/test_bootstrap_lt.log
timeout -k 10 300 tests/cpp/test_bootstrap_eval > gpurun_out/test_bootstrap_eval.log 2>&1 || { tail -40 gpurun_out/test_bootstrap_eval.log; exit 1; }
tail -6 gpurun_out/test_bootstrap_eval.log
timeout -k 10 600 tools/cpp/bench_bootstrap ${1:-16} 16 > gpurun_out/bootstrap.txt 2>&1 || { tail -30 gpurun_out/bootstrap.txt; exit 1; }
cat gpurun_out/bootstrap.txt
