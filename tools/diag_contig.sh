#!/bin/bash
# diagnostic builds of the forward contiguous pass: (a) without twiddle loads, (b) with direct stores
set -e
cd $(dirname $0)/..
SRC=moai-fhe-transformerinference-public_amd/csrc
for v in base notw direct; do
  mkdir -p gpurun_out/diag_$v
  flags=""
  [ $v = notw ] && flags="-DMOAI_DIAG_NO_TW"
  [ $v = direct ] && flags="-DMOAI_DIAG_DIRECT_STORE"
  for f in context ntt elementwise keyswitch; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $flags -c $SRC/$f.hip -o gpurun_out/diag_$v/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/diag_$v/lib.so gpurun_out/diag_$v/*.o
  echo "== $v"
  MOAI_HIP_LIB=$PWD/gpurun_out/diag_$v/lib.so python - <<'PY'
import sys
sys.path.insert(0, ".")
import torch
import __graft_entry__ as g, bench
m = g.load_package()
primes = bench.primes_44x60()
ctx = m.Context(16, primes)
B = 64
data = torch.randint(0, 1 << 59, (B, 2, 44, 65536), dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ev = [m.hip.Event() for _ in range(2)]
for _ in range(2):
    ctx.ntt_forward(data.data_ptr(), B * 2, 44, stream=st)
f = 0
for _ in range(5):
    ev[0].record(st); ctx.ntt_forward(data.data_ptr(), B * 2, 44, stream=st); ev[1].record(st)
    f += ev[1].elapsed_ms_since(ev[0])
print("batch 64 forward: %.3f ms" % (f / 5))
PY
done
