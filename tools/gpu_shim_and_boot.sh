set -e
cd /root/repo
mkdir -p gpurun_out
for t in test_seal_shim test_moai_headers test_bootstrap_lt test_bootstrap_eval; do
  timeout -k 10 400 tests/cpp/$t > gpurun_out/$t.log 2>&1 || { tail -30 gpurun_out/$t.log; exit 1; }
  echo "$t: $(tail -1 gpurun_out/$t.log)"
done
timeout -k 10 600 tools/cpp/bench_bootstrap ${1:-16} 16 ${2:-16} > gpurun_out/bootstrap.txt 2>&1 || { tail -30 gpurun_out/bootstrap.txt; exit 1; }
cat gpurun_out/bootstrap.txt
