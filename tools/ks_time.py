#!/usr/bin/env python3
"""time rotate_vector's core (apply_galois = permute + key switch) at MOAI parameters on the GPU"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
import bench

m = g.load_package()
N = 65536
bits = [51] + [46] * 20 + [51] * 14 + [58]
primes = []
tab = {}
for b in sorted(set(bits)):
    cnt = bits.count(b)
    v = ((1 << b) - 1) // (2 * N) * (2 * N) + 1
    found = []
    while len(found) < cnt:
        if bench.is_prime(v):
            found.append(v)
        v -= 2 * N
    tab[b] = found
for b in bits:
    primes.append(tab[b].pop())
ctx = m.Context(16, primes)
k = len(primes)
dev = torch.device("cuda")
key = torch.randint(0, 1 << 45, (k - 1, 2, k, N), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
for L, B in [(35, 1), (35, 8), (35, 64), (21, 64), (15, 64), (3, 64), (35, 256)]:
    if len(sys.argv) > 1 and B > int(sys.argv[1]):
        continue
    ct = torch.randint(0, 1 << 45, (B, 2, L, N), dtype=torch.int64, device=dev)
    elt = ctx.galois_elt_from_step(1)
    ctx.apply_galois(ct.data_ptr(), L, elt, key.data_ptr(), B, stream=st)
    torch.cuda.synchronize()
    e0, e1 = m.hip.Event(), m.hip.Event()
    reps = 3
    e0.record(st)
    for _ in range(reps):
        ctx.apply_galois(ct.data_ptr(), L, elt, key.data_ptr(), B, stream=st)
    e1.record(st)
    ms = e1.elapsed_ms_since(e0) / reps
    print("L=%2d batch=%3d: %9.3f ms per call, %8.3f ms per ciphertext" % (L, B, ms, ms / B), flush=True)
    del ct
