#!/usr/bin/env python3
"""Does the intermediate of the two-pass NTT stay in the Infinity Cache when a launch pair is kept small?
Times the forward and inverse transform of bench.py's workload (N = 2^16, 44 x 60-bit primes) over 256 polynomials
issued as sub-batches: a sub-batch of c polynomials is c * 44 * 512 KiB between the strided and the contiguous pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as g
import bench

m = g.load_package()
N, K, TOTAL = 65536, 44, 256
primes = bench.primes_44x60()
ctx = m.Context(16, primes)
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
x = torch.randint(0, 1 << 59, (TOTAL, K, N), dtype=torch.int64, device=dev)
psz = K * N * 8
for sub in (256, 64, 16, 8, 4, 2, 1):
    for name, fn in (("forward", ctx.ntt_forward), ("inverse", ctx.ntt_inverse)):
        def run():
            for i in range(0, TOTAL, sub):
                fn(x.data_ptr() + i * psz, sub, K, stream=st)
        run()
        torch.cuda.synchronize()
        e0, e1 = m.hip.Event(), m.hip.Event()
        e0.record(st)
        run(); run()
        e1.record(st)
        ms = e1.elapsed_ms_since(e0) / 2
        gbs = TOTAL * psz * 2 / ms / 1e6
        print("%s  sub-batch %3d polys (%6.1f MiB between passes): %8.3f ms  %7.1f GB/s algorithmic" % (name, sub, sub * psz / 2**20, ms, gbs), flush=True)
    x.random_(0, 1 << 59)
