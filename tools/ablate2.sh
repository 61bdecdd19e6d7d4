#!/bin/bash
# diagnostic: where the NTT passes' time goes.  Builds throw-away copies of the library (results are WRONG
# by construction) and times forward / inverse at batch 64 of the bench shape:
#   full      the product
#   notw      arithmetic kept, twiddles taken from kernel scalars instead of the tables (no table traffic)
#   noarith   twiddles loaded, butterflies replaced by an add/xor (MOAI_ABLATE=1)
#   neither   data movement only
set -e
cd $(dirname $0)/..
SRC=moai-fhe-transformerinference-public_amd/csrc
OUT=gpurun_out/ablate2
mkdir -p $OUT
build() { # name, extra flags, patch twiddles?
  d=$OUT/$1; mkdir -p $d/csrc $d/include; cp $SRC/*.h $SRC/*.cuh $SRC/*.hip $d/csrc/; cp include/moai_hip.h $d/include/
  sed -i 's#"../../include/moai_hip.h"#"../include/moai_hip.h"#' $d/csrc/common.h
  if [ "$3" = "1" ]; then
    sed -i -E 's/Tw t = (tw|twbt)\[.*\];/Tw t; t.w = q >> 1; t.wq = q2 >> 3;/' $d/csrc/ntt_kernels.cuh
  fi
  for f in context ntt elementwise keyswitch encoder; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $2 -c $d/csrc/$f.hip -o $d/$f.o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libmoai.so $d/*.o
}
build full "" 0
build notw "" 1
build noarith "-DMOAI_ABLATE=1" 0
build neither "-DMOAI_ABLATE=1" 1
for v in full notw noarith neither; do
MOAI_HIP_LIB=$PWD/$OUT/$v/libmoai.so python - $v <<'PY'
import sys
sys.path.insert(0, ".")
import torch
import __graft_entry__ as g, bench
m = g.load_package()
ctx = m.Context(16, bench.primes_44x60())
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
print("%-8s batch 64: fwd %.3f ms  inv %.3f ms" % (sys.argv[1], f / 5, i / 5), flush=True)
PY
done
