set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 ${4:-1000} tools/cpp/bench_encoder_layer ${1:-12} ${2:-16} ${3:-48} ${5:-48} > /dev/null 2> gpurun_out/encoder_layer.txt || { tail -30 gpurun_out/encoder_layer.txt; exit 1; }
cat gpurun_out/encoder_layer.txt
