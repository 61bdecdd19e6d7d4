set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 300 tests/cpp/test_bootstrap_eval > gpurun_out/test_bootstrap_eval.log 2>&1 || { tail -30 gpurun_out/test_bootstrap_eval.log; exit 1; }
tail -1 gpurun_out/test_bootstrap_eval.log
timeout -k 10 600 tools/cpp/bench_softmax 16 13 > gpurun_out/softmax.txt 2>&1 || { tail -30 gpurun_out/softmax.txt; exit 1; }
cat gpurun_out/softmax.txt
