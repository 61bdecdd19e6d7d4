set -e
cd /root/repo
timeout -k 10 300 tests/cpp/test_moai_headers > gpurun_out/headers.log 2>&1 || { tail -30 gpurun_out/headers.log; exit 1; }
tail -3 gpurun_out/headers.log
timeout -k 10 300 tests/cpp/test_seal_shim > gpurun_out/shim.log 2>&1 || { tail -30 gpurun_out/shim.log; exit 1; }
tail -2 gpurun_out/shim.log
timeout -k 10 300 tests/cpp/test_bootstrap_lt > gpurun_out/lt.log 2>&1 || { tail -30 gpurun_out/lt.log; exit 1; }
tail -1 gpurun_out/lt.log
timeout -k 10 900 tools/cpp/bench_attention 16 768 > gpurun_out/attention.txt 2>&1 || { tail -20 gpurun_out/attention.txt; exit 1; }
cat gpurun_out/attention.txt
