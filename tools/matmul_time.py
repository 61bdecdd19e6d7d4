#!/usr/bin/env python3
"""time the fused ct-pt matrix product at MOAI's attention shape: X = 768 ciphertexts at chain index 15
(16 primes), W 768 x 64 (single_att_block.hpp:30), N = 2^16; and the same work as separate
multiply_plain + add calls the way the reference issues it."""
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
    v = ((1 << b) - 1) // (2 * N) * (2 * N) + 1
    found = []
    while len(found) < bits.count(b):
        if bench.is_prime(v):
            found.append(v)
        v -= 2 * N
    tab[b] = found
for b in bits:
    primes.append(tab[b].pop())
ctx = m.Context(16, primes)
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
rows, cols, L = 768, 64, 16
x = torch.randint(0, 1 << 45, (rows, 2, L, N), dtype=torch.int64, device=dev)
w = torch.randint(0, 1 << 45, (L, rows, cols), dtype=torch.int64, device=dev)
out = torch.empty((cols, 2, L, N), dtype=torch.int64, device=dev)
res = torch.empty((cols, 2, L - 1, N), dtype=torch.int64, device=dev)
e0, e1, e2 = m.hip.Event(), m.hip.Event(), m.hip.Event()
ctx.ct_pt_matmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), rows, cols, 2, L, stream=st)
torch.cuda.synchronize()
e0.record(st)
ctx.ct_pt_matmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), rows, cols, 2, L, stream=st)
e1.record(st)
ctx.rescale(out.data_ptr(), res.data_ptr(), 2, L, cols, stream=st)
e2.record(st)
t_mm, t_rs = e1.elapsed_ms_since(e0), e2.elapsed_ms_since(e1)
macs = rows * cols * 2 * L * N
print("fused ct-pt matmul 768x64 @ 16 primes: %.2f ms (%.1f G modular MAC/s) + rescale of 64 outputs %.2f ms" % (t_mm, macs / t_mm / 1e6, t_rs))
# the reference's call pattern for ONE output column: 768 x (multiply_plain + add_inplace), one rescale
tmp = torch.empty((2, L, N), dtype=torch.int64, device=dev)
acc = torch.zeros((2, L, N), dtype=torch.int64, device=dev)
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(rows):
    ctx.mul_scalar_rows(x[j].data_ptr(), [12345 + j] * L, tmp.data_ptr(), 2, L, stream=st)
    ctx.add(acc.data_ptr(), tmp.data_ptr(), acc.data_ptr(), 2, L, stream=st)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print("per-op pattern, one column: %.2f ms -> %.1f ms for 64 columns (host-issue bound: %.1f us per call)" % (t1 * 1e3, t1 * 1e3 * cols, t1 / (2 * rows) * 1e6))
