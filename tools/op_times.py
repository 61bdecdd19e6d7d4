#!/usr/bin/env python3
"""Per-primitive GPU timings at MOAI's parameters (N = 2^16, the 36-prime chain), the table of BASELINE.md
section 2 / SURVEY.md section 6 measured on the device: one ciphertext per call (how MOAI calls the
evaluator) and batched."""
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
k = len(primes)
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
key = torch.randint(0, 1 << 45, (k - 1, 2, k, N), dtype=torch.int64, device=dev)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = m.hip.Event(), m.hip.Event()
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    return e1.elapsed_ms_since(e0) / reps


print("%-34s %10s %10s %10s %10s   (ms per ciphertext)" % ("op", "l=35", "l=21", "l=15", "l=3"))
for B in (1, 32):
    rows = {}
    for L in (35, 21, 15, 3):
        ct = torch.randint(0, 1 << 45, (B, 2, L, N), dtype=torch.int64, device=dev)
        ct2 = torch.randint(0, 1 << 45, (B, 2, L, N), dtype=torch.int64, device=dev)
        ct3 = torch.randint(0, 1 << 45, (B, 3, L, N), dtype=torch.int64, device=dev)
        out2 = torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)
        out3 = torch.empty((B, 3, L, N), dtype=torch.int64, device=dev)
        pt = torch.randint(0, 1 << 45, (L, N), dtype=torch.int64, device=dev)
        elt = ctx.galois_elt_from_step(1)
        reps = 20 if L < 20 else 5
        ops = {
            "rotate_vector (1 key switch)": lambda: ctx.apply_galois(ct.data_ptr(), L, elt, key.data_ptr(), B, stream=st),
            "relinearize 3->2": lambda: ctx.relinearize(ct3.data_ptr(), key.data_ptr(), out2.data_ptr(), L, B, stream=st),
            "multiply ct x ct (2x2->3)": lambda: ctx.ct_multiply(ct.data_ptr(), ct2.data_ptr(), out3.data_ptr(), L, B, stream=st),
            "multiply_plain (vector pt)": lambda: ctx.dyadic_mul(ct.data_ptr(), pt.data_ptr(), out2.data_ptr(), 2 * B, 1, L, stream=st),
            "multiply_plain (scalar pt)": lambda: ctx.mul_scalar_rows(ct.data_ptr(), [12345] * L, out2.data_ptr(), 2 * B, L, stream=st),
            "add": lambda: ctx.add(ct.data_ptr(), ct2.data_ptr(), out2.data_ptr(), 2 * B, L, stream=st),
            "INTT + NTT of a size-2 ct": lambda: (ctx.ntt_inverse(ct.data_ptr(), 2 * B, L, stream=st), ctx.ntt_forward(ct.data_ptr(), 2 * B, L, stream=st)),
        }
        if L > 1:
            outr = torch.empty((B, 2, L - 1, N), dtype=torch.int64, device=dev)
            ops["rescale_to_next"] = lambda: ctx.rescale(ct.data_ptr(), outr.data_ptr(), 2, L, B, stream=st)
        for name, fn in ops.items():
            rows.setdefault(name, []).append(timeit(fn, reps) / B)
        del ct, ct2, ct3, out2, out3
    print("-- batch %d" % B)
    for name, v in rows.items():
        print("%-34s " % name + " ".join("%10.4f" % x for x in v), flush=True)
