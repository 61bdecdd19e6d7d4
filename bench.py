#!/usr/bin/env python3
"""bench.py -- the hot-path benchmark of BASELINE.json configs[1]: negacyclic NTT/INTT microbench,
N = 2^16, L = 44 60-bit primes, batch = 256 ciphertexts (2 polynomials each) per GPU.

A "step" is one pass of the path over one batch of synthetic input already resident in HBM:
forward NTT of all 256 x 2 x 44 rows followed by the inverse NTT of the same rows.
Algorithmic bytes: 16 B per coefficient per transform (8 read + 8 written), i.e.
23.62 GB per direction per batch, 47.24 GB per step.  `value` is algorithmic GB/s over the whole
job (all ranks), `roofline` prices the forward transform (its two kernels back to back) against the
8 TB/s HBM peak, measured with HIP events on the stream the kernels run on.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: every rank owns an independent batch (the path shards on the ciphertext index with no
exchange step); torch.distributed is used only for the barrier and the max-over-ranks
time.  The CPU baseline leg runs the oracle (test infrastructure, `kind: port`) on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOGN = 16
N = 1 << LOGN
L = 44
BATCH = 256
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="ciphertexts per GPU (default: the config's 256)")
    ap.add_argument("--split", action="store_true",
                    help="strong scaling: divide ONE batch of --batch ciphertexts over the ranks (shard.shard_range) "
                         "instead of giving every rank its own batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    import __graft_entry__ as g

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    m = g.load_package()  # raises if libmoai_hip.so is missing: no fallback
    primes = primes_44x60()
    ctx = m.Context(LOGN, primes, device=local_rank)

    # weak scaling (default): every rank owns a whole batch.  --split: the ranks share one batch, each taking the
    # contiguous range of ciphertexts shard_range gives it (BASELINE.json north_star: "the 256-ciphertext input
    # batch shards embarrassingly across the 8 GPUs"); still no exchange step.
    B, first_ct = batch_of_rank(args.batch, rank, world, args.split, m.shard)
    if B == 0:
        raise SystemExit("--split with more ranks than ciphertexts leaves rank %d without work" % rank)
    n_poly = B * 2
    # synthetic residues, uniform in [0, q_i) per RNS row, generated on the device
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + (first_ct if args.split else rank))
    data = torch.empty((B, 2, L, N), dtype=torch.int64, device=dev)
    for i, q in enumerate(primes):
        data[:, :, i, :] = torch.randint(0, q, (B, 2, N), dtype=torch.int64, device=dev, generator=gen)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    ptr = data.data_ptr()

    def step():
        ctx.ntt_forward(ptr, n_poly, L, stream=stream)
        ctx.ntt_inverse(ptr, n_poly, L, stream=stream)

    # keep a sample to verify the round trip and parity after the timed region
    sample_before = data[0, 0, :, :].clone()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    m.shard.barrier(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    m.shard.barrier(dist)
    torch.cuda.synchronize()
    elapsed = m.shard.max_over_ranks(time.perf_counter() - t0, dist, dev)

    # round trip must be the identity (bit exact)
    assert torch.equal(data[0, 0, :, :], sample_before), "INTT(NTT(x)) != x"

    # per-direction kernel time with HIP events on the launch stream (outside the timed region)
    ev = [m.hip.Event() for _ in range(3)]
    reps = 5
    fwd_ms, inv_ms = [], []
    for _ in range(reps):
        ev[0].record(stream)
        ctx.ntt_forward(ptr, n_poly, L, stream=stream)
        ev[1].record(stream)
        ctx.ntt_inverse(ptr, n_poly, L, stream=stream)
        ev[2].record(stream)
        fwd_ms.append(ev[1].elapsed_ms_since(ev[0]))
        inv_ms.append(ev[2].elapsed_ms_since(ev[1]))
    fwd = sum(fwd_ms) / reps
    inv = sum(inv_ms) / reps

    coeffs = n_poly * L * N
    bytes_dir = 16.0 * coeffs
    ms_per_step = elapsed / args.steps * 1e3
    # whole job: all ranks' rows over the slowest rank's time (weak: world batches; split: the one batch)
    job_coeffs = m.shard.sum_over_ranks(float(coeffs), dist, dev)
    value = 2 * 16.0 * job_coeffs / (elapsed / args.steps) / 1e9  # GB/s

    out = {
        "metric": "NTT+INTT algorithmic GB/s (N=2^16, L=44x60-bit, batch=256 ciphertexts/GPU) vs HBM peak",
        "value": round(value, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if args.split else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: negacyclic NTT then INTT, N=65536, 44 x 60-bit primes, %s" % (
                "%d ciphertexts x 2 polys split over %d GPUs" % (args.batch, world) if args.split
                else "%d ciphertexts x 2 polys per GPU" % B),
            "batch_per_gpu": B,
            "batch_total": args.batch if args.split else B * world,
            "coeff_modulus": "CoeffModulus::Create(65536, 44 x 60)",
            "bytes_per_coeff_per_transform": 16,
        },
    }
    if rank == 0:
        achieved = bytes_dir / (fwd * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm",
            "kernel": "forward NTT = ntt_fwd_strided<16> + ntt_fwd_contig<16>",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": load_traffic(),
            "fwd_ms": round(fwd, 4),
            "inv_ms": round(inv, 4),
            "inv_achieved": round(bytes_dir / (inv * 1e-3) / 1e9, 1),
            "algorithmic_bytes_per_launch": bytes_dir,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(primes, args.cpu_seconds, sample_before.cpu().numpy().view(np.uint64),
                                              data, ctx, stream)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def batch_of_rank(batch, rank, world, split, shard):
    """(ciphertexts this rank transforms, index of its first one in the job's batch)"""
    if not split:
        return batch, rank * batch
    start, count = shard.shard_range(batch, rank, world)
    return count, start


def primes_44x60():
    """CoeffModulus::Create(65536, 44 x {60}) (SEAL/modulus.cpp:142-183): the 44 largest primes
    = 1 mod 2^17 below 2^60, smallest first.  Product-side generation (no oracle involved)."""
    factor = 2 * N
    v = ((1 << 60) - 1) // factor * factor + 1
    found = []
    while len(found) < L:
        if is_prime(v):
            found.append(v)
        v -= factor
    return found[::-1]


def is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def load_traffic():
    """HBM bytes per forward-NTT launch pair from the committed PMC run (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get("ntt_forward_hbm_bytes_per_launch")
    except Exception:
        return None


def cpu_baseline(primes, seconds, sample_before, data, ctx, stream):
    """Time the oracle (CPU restatement of SEAL's radix-2 Harvey NTT, OpenMP over polynomials like
    MOAI's `#pragma omp parallel for` over ciphertexts) on this host, on a bounded sample of the same
    workload, and use it to check the GPU result of that sample bit for bit."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O  # checker + baseline only

    # torch has already initialised the OpenMP runtime, so the environment variable would come too late
    O.lib().mo_set_threads(host_cores())
    threads = O.lib().mo_max_threads()
    octx = O.Context(LOGN, primes)
    # parity of one polynomial's 44 rows: GPU forward vs oracle forward
    import torch

    x = torch.from_numpy(sample_before.view(np.int64).copy()).to(data.device)
    ctx.ntt_forward(x.data_ptr(), 1, L, stream=stream)
    torch.cuda.synchronize()
    gpu_fwd = x.cpu().numpy().view(np.uint64)
    cpu_fwd = octx.ntt(sample_before.reshape(1, L, N), L)[0]
    assert (gpu_fwd == cpu_fwd).all(), "GPU forward NTT differs from the oracle"

    # timed sample: `cts` ciphertexts (2 x 44 rows each), forward + inverse, repeated until >= seconds
    cts = max(1, threads // 2)
    rng = np.random.default_rng(1)
    buf = O.uniform_rns(rng, primes, (cts * 2,), N)
    import ctypes as C

    t0 = time.perf_counter()
    done = 0
    while True:
        O.lib().mo_batch_ntt(octx.h, O.ptr(buf), cts * 2, L, None, 0)
        O.lib().mo_batch_ntt(octx.h, O.ptr(buf), cts * 2, L, None, 1)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    transforms = done * 2 * cts * 2 * L
    gbs = transforms * N * 16.0 / el / 1e9
    return {
        "value": round(gbs, 3),
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d ciphertexts x 2 polys x 44 rows, forward+inverse, %d repetitions in %.1f s (%.1f us per N=2^16 transform per thread)"
        % (cts, done, el, el * threads / transforms * 1e6),
    }


if __name__ == "__main__":
    main()
