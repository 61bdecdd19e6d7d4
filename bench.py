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
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end slice (bootstrap + attention head at MOAI's parameters)")
    ap.add_argument("--e2e-pack", type=int, default=48, help="ciphertexts per packed bootstrap in the end-to-end slice")
    ap.add_argument("--e2e-timeout", type=float, default=540.0)
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
        out["roofline"].update(second_roofs(fwd, coeffs, out["roofline"]["traffic"]))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(primes, args.cpu_seconds, sample_before.cpu().numpy().view(np.uint64),
                                              data, ctx, stream)
        if world == 1 and not args.no_e2e:
            # the metric's first half (ms per encrypted input): a bounded slice at MOAI's parameters in a CHILD process
            # (never a re-exec of this one), after this process has given its device memory back
            sample_host = None
            del data, sample_before
            ctx.close()
            torch.cuda.empty_cache()
            try:
                out["e2e"] = end_to_end_slice(args, None if args.no_cpu_baseline else host_cores())
            except Exception as e:  # the NTT line must come out whatever happens to the slice
                out["e2e"] = {"skipped": "end_to_end_slice raised %r" % (e,)}
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


SHOUP_BFLY_PER_S = 1.716e12   # 64-bit Harvey/Shoup butterflies per second, whole chip, no memory access (profiles/r01_valu_bench.txt)
COPY_RATE_GBS = 5280.0        # read+write rate of an in-place 32 KiB-tile kernel on this part (same file); HBM3E spec 8000


def second_roofs(fwd_ms, coeffs, traffic_bytes):
    """The forward transform against its other roofs: the integer pipes (one butterfly per coefficient pair and stage)
    and the HBM traffic the two-launch design really moves (PMC), at the copy rate the part sustains."""
    bfly = coeffs / 2 * LOGN
    valu_floor = bfly / SHOUP_BFLY_PER_S * 1e3
    out = {
        "valu": {"achieved": round(bfly / (fwd_ms * 1e-3) / 1e9, 1), "peak": SHOUP_BFLY_PER_S / 1e9, "unit": "Gbutterfly/s",
                 "frac": round(valu_floor / fwd_ms, 4), "butterflies_per_launch": bfly,
                 "source": "profiles/r01_valu_bench.txt (64-bit Shoup butterfly loop, no memory traffic)"},
        "floors_ms": {"hbm_algorithmic_at_spec_peak": round(16.0 * coeffs / HBM_PEAK_GBS / 1e6, 3), "valu": round(valu_floor, 3)},
    }
    if traffic_bytes:
        two_pass = traffic_bytes / COPY_RATE_GBS / 1e6
        out["floors_ms"]["hbm_measured_traffic_at_copy_rate"] = round(two_pass, 3)
        out["binding_roof"] = ("hbm: the PMC traffic of the two launches at the %.0f GB/s a read-modify-write kernel sustains "
                               "needs %.2f ms, the integer pipes alone %.2f ms" % (COPY_RATE_GBS, two_pass, valu_floor))
        out["frac_of_binding_floor"] = round(max(two_pass, valu_floor) / fwd_ms, 4)
    return out


def end_to_end_slice(args, cores):
    """Runs tools/cpp/bench_e2e (built by __graft_entry__.build(); includes MOAI's own headers, so it is built where the
    reference checkout exists and travels prebuilt) and prices the operations it counted on the CPU oracle."""
    import subprocess

    exe = os.path.join(ROOT, "tools", "cpp", "bench_e2e")
    if not os.path.exists(exe):
        return {"skipped": "tools/cpp/bench_e2e is not built on this machine"}
    t0 = time.perf_counter()
    try:
        r = subprocess.run([exe, str(args.e2e_pack), str(host_cores())], cwd=os.path.dirname(exe), capture_output=True, text=True,
                           timeout=args.e2e_timeout)
    except subprocess.TimeoutExpired:
        return {"skipped": "bench_e2e exceeded %.0f s" % args.e2e_timeout}
    line = [l for l in r.stdout.splitlines() if l.startswith("E2E_JSON ")]
    if r.returncode != 0 or not line:
        return {"skipped": "bench_e2e failed (exit %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    child = json.loads(line[-1][len("E2E_JSON "):])
    notes = [l.strip()[:300] for l in r.stdout.splitlines()
             if l.startswith(("setup", "bootstrap_3:", "  [device memory", "single_att_block", "Compute Q, K, V", "768 input"))]
    ops_boot = child.pop("ops_bootstrap_pack")
    ops_head = child.pop("ops_head")
    stages = layer_stage_profile()
    rest_s = stages["rest_s"] if stages else None
    out = {
        "workload": "MOAI parameters (N=65536, 36-prime chain {51,46x20,51x14,58}, logn=15, K=25, degree-59 cosine): bootstrap_3 on a pack "
                    "of %d, and one attention head through MOAI's own single_att_block.hpp / softmax.hpp (unchanged) on 768 input "
                    "ciphertexts = 256 packed inputs x 128 tokens, synthetic weights" % child["pack"],
        "bootstrap_ms": child["bootstrap_ms_packed"],
        "bootstrap_ms_moai_call_pattern": child["bootstrap_ms_moai_calls"],
        "bootstrap_max_error": child["bootstrap_max_error"],
        "bootstrap_chain_index_after": child["bootstrap_chain_index_after"],
        "head_s": child["head_s"],
        "head_max_error": child["head_max_error"],
        "head_max_error_vs_exact_softmax": child["head_max_error_vs_exact_softmax"],
        "child_wall_s": round(time.perf_counter() - t0, 1),
        "child_setup_s": child["setup_s"],
        "child_log": notes,
    }
    # per layer: 12 heads + 4 x 768 bootstraps + the stages the child does not run (self-output, LayerNorm x2, the
    # feed-forward products, GELU), taken from the committed whole-layer run of this round
    measured_s = 12 * out["head_s"] + 3072 * out["bootstrap_ms"] * 1e-3
    out["layer_measured_part_s"] = round(measured_s, 2)
    if rest_s is not None:
        out["layer_rest_s"] = {"value": rest_s, "source": stages["source"]}
        out["projected_ms_per_input"] = round(12 * (measured_s + rest_s) / 256 * 1e3, 1)
        # the same with the attention of the committed whole-layer run (fused products instead of MOAI's per-ciphertext loops)
        if stages.get("attention_12_heads_s"):
            fused = stages["attention_12_heads_s"] + 3072 * out["bootstrap_ms"] * 1e-3 + rest_s
            out["projected_ms_per_input_fused_attention"] = {"value": round(12 * fused / 256 * 1e3, 1),
                                                             "attention_12_heads_s": stages["attention_12_heads_s"], "source": stages["source"]}
    if cores:
        try:
            out["cpu_baseline"] = price_on_cpu(ops_boot, ops_head, child["pack"], cores)
            cb = out["cpu_baseline"]
            cpu_measured_s = 12 * cb["head_s"] + 3072 * cb["bootstrap_ms"] * 1e-3
            cb["layer_measured_part_s"] = round(cpu_measured_s, 1)
            out["speedup_vs_cpu_measured_part"] = round(cpu_measured_s / measured_s, 1)
        except Exception as e:  # the checker must never sink the bench line
            out["cpu_baseline"] = {"skipped": repr(e)[:200]}
    return out


def layer_stage_profile():
    p = os.path.join(ROOT, "profiles", "r02_encoder_layer_stages.json")
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:
        return None


def moai_primes():
    """CoeffModulus::Create(65536, {51, 46 x 20, 51 x 14, 58}) (include/test/test_full_scheme.hpp:356-378)"""
    bits = [51] + [46] * 20 + [51] * 14 + [58]
    tab = {}
    for b in sorted(set(bits)):
        v = ((1 << b) - 1) // (2 * N) * (2 * N) + 1
        found = []
        while len(found) < bits.count(b):
            if is_prime(v):
                found.append(v)
            v -= 2 * N
        tab[b] = found
    return [tab[b].pop() for b in bits]


def price_on_cpu(ops_boot, ops_head, pack, cores):
    """CPU baseline of the end-to-end slice: the operations the GPU run was asked to perform (library census), each priced
    with the oracle's time for that primitive on this host -- key switch, rescale, ct x ct product, plaintext product,
    addition, one NTT row, measured at two or three levels and interpolated (a key switch is l^2+3l+2 transforms, the
    others are linear in l) -- summed, and divided by the cores as MOAI's OpenMP loops over ciphertexts divide it."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as O  # checker + baseline only

    primes = moai_primes()
    k = len(primes)
    octx = O.Context(LOGN, primes)
    rng = np.random.default_rng(7)
    O.lib().mo_set_threads(1)
    t_begin = time.perf_counter()

    def timed_pool(fn, n):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(fn, range(n)))
        return (time.perf_counter() - t0) * cores / n  # thread-seconds per call

    key = O.uniform_rns(rng, primes, (k - 1, 2), N)
    ks, lin = {}, {}
    for L in (35, 21, 9):
        ct = O.uniform_rns(rng, primes[:L], (2,), N)
        elt = O.galois_elt_from_step(LOGN, 1)
        ks[L] = timed_pool(lambda i: octx.apply_galois(ct, L, elt, key), cores)
    for L in (35, 9):
        ct = O.uniform_rns(rng, primes[:L], (2,), N)
        pt = O.uniform_rns(rng, primes[:L], (), N)
        lin[L] = {
            "rescale": timed_pool(lambda i: octx.rescale(ct, 2, L), cores),
            "mul": timed_pool(lambda i: octx.multiply(ct, ct, L), cores),
            "mulplain": timed_pool(lambda i: octx.multiply_plain(ct, 2, L, pt), cores),
            "add": timed_pool(lambda i: octx.add(ct, ct, 2, L), cores),
        }
    rows = O.uniform_rns(rng, primes[:8], (cores,), N)
    t0 = time.perf_counter()
    O.lib().mo_set_threads(cores)
    O.lib().mo_batch_ntt(octx.h, O.ptr(rows), cores, 8, None, 0)
    t_ntt_row = (time.perf_counter() - t0) * cores / (cores * 8)

    def t_ks(L):
        # quadratic through the three measured levels in x = L^2 + 3L + 2 (the transform count), linear in between
        x = lambda l: l * l + 3 * l + 2
        pts = sorted(ks)
        lo = max([p for p in pts if p <= L] or [pts[0]])
        hi = min([p for p in pts if p >= L] or [pts[-1]])
        if lo == hi:
            return ks[lo] * x(L) / x(lo)
        return ks[lo] + (ks[hi] - ks[lo]) * (x(L) - x(lo)) / (x(hi) - x(lo))

    def t_lin(name, L):
        a, b = lin[9][name], lin[35][name]
        return max(a + (b - a) * (L - 9) / 26.0, a * L / 9.0 if L < 9 else 0.0)

    def price(ops):
        total, by = 0.0, {}
        for name, L, units in ops:
            if name in ("apply_galois_to", "switch_key", "relinearize"):
                c = units * t_ks(L)
            elif name == "rescale":
                c = units / 2 * t_lin("rescale", L)
            elif name == "mul_scalar_rescale":
                c = units / 2 * (t_lin("rescale", L) + t_lin("mulplain", L))
            elif name in ("ct_multiply", "ct_square"):
                c = units * t_lin("mul", L)
            elif name == "ct_dot":
                c = units * (t_lin("mul", L) + 1.5 * t_lin("add", L))
            elif name in ("dyadic_mul", "mul_scalar_rows"):
                c = units / 2 * t_lin("mulplain", L)
            elif name in ("ct_pt_dot", "ct_pt_matmul"):
                c = units / 2 * (t_lin("mulplain", L) + t_lin("add", L))
            elif name in ("add", "sub", "negate", "add_scalar_rows", "galois_permute", "mod_drop"):
                c = units / 2 * t_lin("add", L)
            elif name in ("ntt_forward", "ntt_inverse"):
                c = units * L * t_ntt_row
            elif name == "modraise":
                c = units * 2 * (L + 1) * t_ntt_row
            else:  # ckks_encode*: the reference encodes a plaintext per product; not priced (conservative for the CPU)
                c = 0.0
            by[name] = by.get(name, 0.0) + c
            total += c
        return total, by

    boot_thread_s, boot_by = price(ops_boot)
    head_thread_s, head_by = price(ops_head)
    top = lambda d: {n: round(v, 2) for n, v in sorted(d.items(), key=lambda kv: -kv[1])[:4]}
    return {
        "kind": "port",
        "cores": cores,
        "bootstrap_ms": round(boot_thread_s / pack / cores * 1e3, 1),
        "head_s": round(head_thread_s / cores, 1),
        "thread_seconds": {"bootstrap_pack": round(boot_thread_s, 1), "head": round(head_thread_s, 1),
                           "largest_terms_bootstrap": top(boot_by), "largest_terms_head": top(head_by)},
        "primitive_thread_seconds": {"key_switch": {str(L): round(v, 3) for L, v in ks.items()},
                                     "l35": {n: round(v, 4) for n, v in lin[35].items()}, "ntt_row": round(t_ntt_row, 6)},
        "sample": "oracle primitives timed for %.1f s on %d threads (key switch at l=35/21/9, rescale / products / add at l=35/9, one "
                  "NTT row), multiplied by the operation counts of the GPU run and divided by the cores; plaintext encodes not priced"
                  % (time.perf_counter() - t_begin, cores),
    }


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
