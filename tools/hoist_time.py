#!/usr/bin/env python3
"""R rotations of the same ciphertexts: one hoisted call against R separate apply_galois_to calls (MOAI parameters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
import bench

m = g.load_package()
N = 65536
primes = bench.moai_primes()
ctx = m.Context(16, primes)
k = len(primes)
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
L = int(sys.argv[1]) if len(sys.argv) > 1 else 35
B = int(sys.argv[2]) if len(sys.argv) > 2 else 48
R = int(sys.argv[3]) if len(sys.argv) > 3 else 7
keys = [torch.randint(0, 1 << 45, (k - 1, 2, k, N), dtype=torch.int64, device=dev) for _ in range(R)]
steps = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10][:R]
elts = [ctx.galois_elt_from_step(s) for s in steps]
corrs = [ctx.hoist_correction(kk.data_ptr(), e, L, stream=st) for kk, e in zip(keys, elts)]
ct = torch.randint(1, 1 << 45, (B, 2, L, N), dtype=torch.int64, device=dev)
out = torch.empty((R, B, 2, L, N), dtype=torch.int64, device=dev)
kp = [kk.data_ptr() for kk in keys]
def hoisted():
    return ctx.apply_galois_hoisted(ct.data_ptr(), out.data_ptr(), L, elts, kp, corrs, B, stream=st)
def separate():
    for r in range(R):
        ctx.apply_galois_to(ct.data_ptr(), out[r].data_ptr(), L, elts[r], kp[r], B, stream=st)
for name, fn in (("hoisted", hoisted), ("separate", separate), ("hoisted", hoisted), ("separate", separate)):
    fb = fn()
    torch.cuda.synchronize()
    e0, e1 = m.hip.Event(), m.hip.Event()
    e0.record(st)
    fn(); fn()
    e1.record(st)
    ms = e1.elapsed_ms_since(e0) / 2
    print("l=%d batch=%d R=%d %-9s: %8.2f ms per call = %.3f ms per rotation and ciphertext%s" % (L, B, R, name, ms, ms / (R * B), " (FELL BACK)" if fb else ""), flush=True)
