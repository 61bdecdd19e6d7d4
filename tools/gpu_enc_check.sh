set -e
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/enc_test.log 2>&1 || { tail -40 gpurun_out/enc_test.log; exit 1; }
tail -3 gpurun_out/enc_test.log
timeout -k 10 300 python tools/encode_time.py --cpu > gpurun_out/encode_time.txt 2>&1 || { tail -20 gpurun_out/encode_time.txt; exit 1; }
cat gpurun_out/encode_time.txt
timeout -k 10 300 tests/cpp/test_seal_shim > gpurun_out/shim.log 2>&1 || { tail -20 gpurun_out/shim.log; exit 1; }
tail -3 gpurun_out/shim.log
