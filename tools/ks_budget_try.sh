cd /root/repo
for mb in 2048 4096 8192 16384 45000; do
  echo -n "MOAI_KS_TMP_MB=$mb: "
  MOAI_KS_TMP_MB=$mb timeout -k 10 200 python tools/ks_time.py --only 35 64 2>&1 | grep "L=35"
done
echo -n "B=256 MOAI_KS_TMP_MB=16384: "; MOAI_KS_TMP_MB=16384 timeout -k 10 200 python tools/ks_time.py --only 35 256 2>&1 | grep "L=35"
echo -n "B=256 default: "; timeout -k 10 200 python tools/ks_time.py --only 35 256 2>&1 | grep "L=35"
