cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for mb in 8192 32768; do
  echo "== MOAI_KS_TMP_MB=$mb"
  MOAI_KS_TMP_MB=$mb timeout -k 10 400 tools/cpp/bench_bootstrap 48 16 0 2>&1 | grep "packed\|per bootstrap"
done
