cd /root/repo
for w in 150 500 1500 4000; do
echo "== window $w us"
MOAI_SHIM_COMBINE_US=$w timeout -k 10 600 tools/cpp/bench_bootstrap_lt 32 16 2>&1 | tail -2 | head -1
MOAI_SHIM_COMBINE_US=$w timeout -k 10 900 tools/cpp/bench_attention 16 768 2>&1 | grep "Q K^T (col\|softmax(QK\|gelu_v2 on 16"
done
