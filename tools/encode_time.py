#!/usr/bin/env python3
"""CKKSEncoder::encode at MOAI's parameters (N = 2^16, 36-prime chain): device encoder per vector (single
call and batched) against the CPU oracle's restatement on one thread (the reference encodes on the
calling thread; BASELINE.md section 2 quotes 51 ms at l = 21)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import __graft_entry__ as g
import oracle as O

m = g.load_package()
N = 65536
bits = [51] + [46] * 20 + [51] * 14 + [58]
primes = O.coeff_modulus_create(N, bits)
ctx = m.Context(16, primes)
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(0)
do_cpu = "--cpu" in sys.argv
if do_cpu:
    octx = O.Context(16, primes)
    enc = O.CkksEncoder(octx)

print("%-8s %-8s %14s %14s %14s" % ("level", "batch", "GPU ms/vector", "CPU ms/vector", "bit-identical"))
cases = [(L, B) for L in (35, 21, 15) for B in (1, 64)]
if "--big" in sys.argv:
    cases = [(13, 3072), (2, 768)]  # the column batches of MOAI's masked matrix products (final, self-output)
for L, B in cases:
    if True:
        vals = rng.normal(size=(B, N // 2))
        dv = torch.from_numpy(vals).to(dev)
        out = torch.empty((B, L, N), dtype=torch.int64, device=dev)
        mx = torch.empty(B, dtype=torch.float64, device=dev)

        def run():
            m.hip._check(m.hip.lib().moai_ckks_encode(ctx.h, dv.data_ptr(), 0, N // 2, B, out.data_ptr(), L, None,
                                                      2.0**46, mx.data_ptr(), st))
        run()
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = m.hip.Event(), m.hip.Event()
        e0.record(st)
        for _ in range(reps):
            run()
        e1.record(st)
        gpu_ms = e1.elapsed_ms_since(e0) / reps / B
        cpu_ms, same = float("nan"), "-"
        if do_cpu and B == 1:
            t0 = time.perf_counter()
            want = enc.encode(vals[0], L, 2.0**46)
            cpu_ms = (time.perf_counter() - t0) * 1e3
            same = str(bool((out[0].cpu().numpy().view(np.uint64) == want).all()))
        print("%-8d %-8d %14.4f %14.2f %14s" % (L, B, gpu_ms, cpu_ms, same), flush=True)
