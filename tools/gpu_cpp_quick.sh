set -e
cd /root/repo
for t in test_seal_shim test_moai_headers test_bootstrap_lt; do
  timeout -k 10 300 tests/cpp/$t > gpurun_out/$t.log 2>&1 || { tail -30 gpurun_out/$t.log; exit 1; }
  tail -1 gpurun_out/$t.log
done
timeout -k 10 600 tools/cpp/bench_bootstrap_lt 32 16 2>&1 | tail -3
timeout -k 10 900 tools/cpp/bench_attention 16 768 2>&1 | grep "self-output\|bias"
