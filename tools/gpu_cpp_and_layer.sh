set -e
cd /root/repo
for t in test_seal_shim test_moai_headers test_bootstrap_lt; do
  timeout -k 10 300 tests/cpp/$t > gpurun_out/$t.log 2>&1 || { tail -30 gpurun_out/$t.log; exit 1; }
  tail -1 gpurun_out/$t.log
done
timeout -k 10 900 tools/cpp/bench_layer 16 > gpurun_out/layer_full.txt 2>&1 || { grep -a "LayerNorm\|intermediate\|GELU\|final X\|feed-forward\|rror" gpurun_out/layer_full.txt | cut -c1-300; exit 1; }
grep -a "LayerNorm\|intermediate\|GELU\|final X\|feed-forward" gpurun_out/layer_full.txt | cut -c1-400 > gpurun_out/layer.txt
cat gpurun_out/layer.txt
