#!/bin/bash
# diagnostic: build a copy of the library whose butterflies do no modular arithmetic (MOAI_ABLATE=1)
# and time the NTT passes with it: what remains is the data movement of each kernel.
set -e
cd $(dirname $0)/..
mkdir -p gpurun_out/ablate
SRC=moai-fhe-transformerinference-public_amd/csrc
for f in context ntt elementwise keyswitch; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DMOAI_ABLATE=1 -c $SRC/$f.hip -o gpurun_out/ablate/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/ablate/libmoai_ablate1.so gpurun_out/ablate/*.o
echo "== ablate=1 (no arithmetic), 2-pass"
MOAI_HIP_LIB=$PWD/gpurun_out/ablate/libmoai_ablate1.so MOAI_NTT_COOP=0 python - <<'PY'
import sys, os, time
sys.path.insert(0, ".")
import torch, numpy as np
import __graft_entry__ as g, bench
m = g.load_package()
primes = bench.primes_44x60()
ctx = m.Context(16, primes)
B = 64
data = torch.randint(0, 1 << 59, (B, 2, 44, 65536), dtype=torch.int64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ev = [m.hip.Event() for _ in range(3)]
for _ in range(2):
    ctx.ntt_forward(data.data_ptr(), B * 2, 44, stream=st); ctx.ntt_inverse(data.data_ptr(), B * 2, 44, stream=st)
f = i = 0
for _ in range(5):
    ev[0].record(st); ctx.ntt_forward(data.data_ptr(), B * 2, 44, stream=st); ev[1].record(st)
    ctx.ntt_inverse(data.data_ptr(), B * 2, 44, stream=st); ev[2].record(st)
    f += ev[1].elapsed_ms_since(ev[0]); i += ev[2].elapsed_ms_since(ev[1])
gb = B * 2 * 44 * 65536 * 16 / 1e9
print("batch 64: fwd %.3f ms inv %.3f ms -> %.0f / %.0f GB/s algorithmic (x2 = physical)" % (f / 5, i / 5, gb / (f / 5e3), gb / (i / 5e3)))
PY
