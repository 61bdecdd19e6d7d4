set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/parity.log 2>&1 || { tail -40 gpurun_out/parity.log; exit 1; }
tail -2 gpurun_out/parity.log
for t in test_seal_shim test_bootstrap_eval; do
  timeout -k 10 400 tests/cpp/$t > gpurun_out/$t.log 2>&1 || { tail -30 gpurun_out/$t.log; exit 1; }
  echo "$t: $(tail -1 gpurun_out/$t.log)"
done
timeout -k 10 600 tools/cpp/bench_bootstrap 16 16 16 > gpurun_out/bootstrap_fp.txt 2>&1 || { tail -30 gpurun_out/bootstrap_fp.txt; exit 1; }
cat gpurun_out/bootstrap_fp.txt
MOAI_MD_FP_MIN_ROWS=1000000000 timeout -k 10 600 tools/cpp/bench_bootstrap 16 16 16 > gpurun_out/bootstrap_int.txt 2>&1 || { tail -30 gpurun_out/bootstrap_int.txt; exit 1; }
cat gpurun_out/bootstrap_int.txt
