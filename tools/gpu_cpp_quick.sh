set -e
cd /root/repo
timeout -k 10 300 tests/cpp/test_moai_headers > gpurun_out/test_moai_headers.log 2>&1 || { tail -30 gpurun_out/test_moai_headers.log; exit 1; }
tail -1 gpurun_out/test_moai_headers.log
timeout -k 10 900 tools/cpp/bench_attention 16 768 > gpurun_out/attention.txt 2>&1 || { tail -20 gpurun_out/attention.txt; exit 1; }
cat gpurun_out/attention.txt
