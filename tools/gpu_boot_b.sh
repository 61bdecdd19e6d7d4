set -e
cd /root/repo
mkdir -p gpurun_out
timeout -k 10 600 tools/cpp/bench_bootstrap ${1:-48} 16 0 > gpurun_out/bootstrap_b.txt 2>&1 || { tail -30 gpurun_out/bootstrap_b.txt; exit 1; }
cat gpurun_out/bootstrap_b.txt
