set -e
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_encoder.py -x -q > gpurun_out/parity_fp.log 2>&1 || { tail -40 gpurun_out/parity_fp.log; exit 1; }
tail -2 gpurun_out/parity_fp.log
timeout -k 10 300 python tools/ks_time.py 64 > gpurun_out/ks_time_fp.txt 2>&1 || { tail -20 gpurun_out/ks_time_fp.txt; exit 1; }
cat gpurun_out/ks_time_fp.txt
MOAI_NTT_FP=0 timeout -k 10 300 python tools/ks_time.py 64 > gpurun_out/ks_time_int.txt 2>&1 || { tail -20 gpurun_out/ks_time_int.txt; exit 1; }
echo "--- MOAI_NTT_FP=0"; cat gpurun_out/ks_time_int.txt
