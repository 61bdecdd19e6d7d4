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
cases = [(35, 1), (35, 8), (35, 64), (21, 64), (15, 64), (3, 64), (35, 256)]
if "--only" in sys.argv:
    i = sys.argv.index("--only")
    cases = [(int(sys.argv[i + 1]), int(sys.argv[i + 2]))]
for L, B in cases:
    if "--only" not in sys.argv and len(sys.argv) > 1 and B > int(sys.argv[1]):
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


# one ciphertext per call replayed from a hipGraph (captured through torch's capture API)
if "--graph" in sys.argv:
    for L in (35, 21, 3):
        ct = torch.randint(0, 1 << 45, (1, 2, L, N), dtype=torch.int64, device=dev)
        elt = ctx.galois_elt_from_step(1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ctx.apply_galois(ct.data_ptr(), L, elt, key.data_ptr(), 1, stream=side.cuda_stream)
        side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            ctx.apply_galois(ct.data_ptr(), L, elt, key.data_ptr(), 1, stream=torch.cuda.current_stream().cuda_stream)
        gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            gr.replay()
        torch.cuda.synchronize()
        print("L=%2d batch=  1 replayed from a hipGraph: %9.3f ms per call" % (L, (time.perf_counter() - t0) / 20 * 1e3), flush=True)

# CPU baseline for the same operation: the oracle's restatement of SEAL's switch_key_inplace
# (SEAL/evaluator.cpp:2724-3020) at MOAI parameters, OpenMP over ciphertexts like MOAI's loops
if "--cpu" in sys.argv:
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle as O
    O.lib().mo_set_threads(bench.host_cores())
    octx = O.Context(16, primes)
    threads = O.lib().mo_max_threads()
    rng = np.random.default_rng(0)
    hkey = key.cpu().numpy().view(np.uint64)
    for L in (35, 15):
        B = threads
        cts = O.uniform_rns(rng, primes[:L], (B, 2), N)
        tg = O.uniform_rns(rng, primes[:L], (B,), N)
        t0 = time.perf_counter()
        out = octx.batch_switch_key(cts, tg, hkey, L, B)
        el = time.perf_counter() - t0
        print("CPU oracle switch_key L=%2d: %d ciphertexts on %d threads in %.2f s -> %.1f ms per ciphertext (%.1f ms single-thread-equivalent)"
              % (L, B, threads, el, el / B * 1e3, el * 1e3), flush=True)
        # parity of ciphertext 0 against the GPU
        dct = torch.from_numpy(cts[0:1].view(np.int64).copy()).to(dev)
        dtg = torch.from_numpy(tg[0:1].view(np.int64).copy()).to(dev)
        ctx.switch_key(dct.data_ptr(), dtg.data_ptr(), key.data_ptr(), L, 1, stream=st)
        torch.cuda.synchronize()
        assert (dct.cpu().numpy().view(np.uint64) == out[0:1]).all(), "GPU key switch differs from the oracle"
        print("   GPU result of ciphertext 0 is bit-identical to the oracle")
