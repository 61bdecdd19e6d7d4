#!/usr/bin/env python3
"""Does the key switch's intermediate stay in the Infinity Cache when a launch pair is kept small?
Times rotate_vector's core at MOAI's top level (l = 35) over 64 ciphertexts issued as sub-batches, under scratch
budgets that make the digits in flight per launch pair fit (or not fit) the 256 MiB last-level cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
import bench

m = g.load_package()
N = 65536
bits = [51] + [46] * 20 + [51] * 14 + [58]
tab, primes = {}, []
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
L = int(sys.argv[1]) if len(sys.argv) > 1 else 35
TOTAL = 64
ct = torch.randint(0, 1 << 45, (TOTAL, 2, L, N), dtype=torch.int64, device=dev)
elt = ctx.galois_elt_from_step(1)
ctsz = 2 * L * N * 8
for sub, mb in [(64, 8192), (64, 32768), (16, 8192), (8, 8192), (8, 150), (4, 8192), (4, 80), (4, 160), (2, 8192), (2, 40), (2, 80), (1, 8192), (1, 20), (1, 40)]:
    m.hip.set_tuning("MOAI_KS_TMP_MB", mb)
    def run():
        for i in range(0, TOTAL, sub):
            ctx.apply_galois(ct.data_ptr() + i * ctsz, L, elt, key.data_ptr(), sub, stream=st)
    run()
    torch.cuda.synchronize()
    e0, e1 = m.hip.Event(), m.hip.Event()
    e0.record(st)
    run(); run()
    e1.record(st)
    ms = e1.elapsed_ms_since(e0) / 2
    print("l=%d  sub-batch %2d  scratch budget %5d MiB : %8.3f ms per ciphertext" % (L, sub, mb, ms / TOTAL), flush=True)
