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
    ap.add_argument("--e2e-timeout", type=float, default=200.0, help="limit of the bootstrap / attention-head / feed-forward-slice child (s)")
    ap.add_argument("--no-layer", action="store_true", help="skip the whole-encoder-layer child of the end-to-end slice (about 210 s)")
    ap.add_argument("--layer-timeout", type=float, default=330.0)
    ap.add_argument("--no-keyswitch", action="store_true", help="skip the configs[2] key-switch region")
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
            # which roof binds is a measured statement (profiles/r03_ntt_pmc_summary.json: the vector ALUs are busy most of the
            # kernels' time, the butterfly IS the kernel); `achieved` / `peak` / `frac` stay the metric's own terms, algorithmic
            # GB/s against the HBM peak
            "bound": "valu",
            "bound_evidence": NTT_BOUND_EVIDENCE,
            "kernel": "forward NTT = ntt_fwd_strided<16,-2> + ntt_fwd_contig<16,-2> (M_LAZY8 integer butterflies, 60-bit primes)",
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
        if world == 1:
            try:
                out["ntt_moai_chain"] = ntt_moai_chain_region(m, args, torch, np, dev, stream, not args.no_cpu_baseline)
            except Exception as e:
                out["ntt_moai_chain"] = {"skipped": "ntt_moai_chain_region raised %r" % (e,)}
        if world == 1 and not args.no_keyswitch:
            try:
                out["keyswitch"] = keyswitch_region(m, args, torch, np, dev, stream, None if args.no_cpu_baseline else host_cores())
            except Exception as e:  # the NTT line must come out whatever happens here
                out["keyswitch"] = {"skipped": "keyswitch_region raised %r" % (e,)}
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
    # N > 1, the metric's first half: one independent packed batch (256 inputs, its own keys) per GPU -- replicas only, no
    # collective.  Every rank frees its device memory and runs the whole-layer child bound to its own GPU.
    if world > 1 and not args.no_e2e and not args.no_layer:
        del data, sample_before
        ctx.close()
        torch.cuda.empty_cache()
        layer_s, why = replica_layer(args, m.shard.replica_env(local_rank, os.environ))
        agg = m.shard.aggregate_replicas(layer_s, 256, 12, dist, dev)
        if rank == 0:
            agg["ms_per_input"] = round(agg["ms_per_input"], 1) if agg["ms_per_input"] else None
            agg["inputs_per_s"] = round(agg["inputs_per_s"], 4)
            agg["what"] = ("one whole encoder layer (tools/cpp/bench_encoder_layer, fused callers) measured on every GPU at the same time, "
                           "each on its own packed batch of 256 inputs; 12 layers = 12 x one layer; inputs per second summed over the "
                           "replicas; no data-path collective")
            if why:
                agg["rank0_replica_problem"] = why
            out["e2e"] = agg
    if rank == 0:
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
    """HBM bytes per forward-NTT launch pair from the committed PMC run (profiles/pmc_traffic.json: separate FETCH_SIZE /
    WRITE_SIZE passes over this same command, corrected as the microarchitecture guide prescribes; the file names the commit
    it was collected at), or None."""
    return (load_json(os.path.join(ROOT, "profiles", "pmc_traffic.json")) or {}).get("ntt_forward_hbm_bytes_per_launch")


SHOUP_BFLY_PER_S = 1.716e12   # the exact 64-bit Harvey/Shoup butterfly (31 VALU instructions; 60/61-bit primes' fallback)
LAZY8_BFLY_PER_S = 1.887e12   # the M_LAZY8 butterfly the bench's kernels run (26 VALU instructions)
COPY_RATE_GBS = 5280.0        # read+write rate of an in-place 32 KiB-tile kernel on this part (same file); HBM3E spec 8000
NTT_BOUND_EVIDENCE = ("profiles/r03_ntt_pmc_summary.json: vector ALUs busy for most of both kernels' cycles, ~26 VALU instructions per "
                      "butterfly; HBM traffic 2.0 x algorithmic (two launches) would need less time than the arithmetic")


def second_roofs(fwd_ms, coeffs, traffic_bytes):
    """The forward transform against its other roofs: the integer pipes (one butterfly per coefficient pair and stage)
    and the HBM traffic the two-launch design really moves (PMC), at the copy rate the part sustains."""
    bfly = coeffs / 2 * LOGN
    valu_floor = bfly / LAZY8_BFLY_PER_S * 1e3
    out = {
        "valu": {"achieved": round(bfly / (fwd_ms * 1e-3) / 1e9, 1), "peak": LAZY8_BFLY_PER_S / 1e9, "unit": "Gbutterfly/s",
                 "frac": round(valu_floor / fwd_ms, 4), "butterflies_per_launch": bfly,
                 "source": "profiles/r03_valu_bench.txt (a loop of M_LAZY8 butterflies, no memory traffic)"},
        "floors_ms": {"hbm_algorithmic_at_spec_peak": round(16.0 * coeffs / HBM_PEAK_GBS / 1e6, 3), "valu": round(valu_floor, 3)},
    }
    if traffic_bytes:
        two_pass = traffic_bytes / COPY_RATE_GBS / 1e6
        out["floors_ms"]["hbm_measured_traffic_at_copy_rate"] = round(two_pass, 3)
        out["ceiling"] = ("max(valu floor %.2f ms, two-pass traffic floor %.2f ms = the PMC-measured bytes at the %.0f GB/s a "
                          "read-modify-write kernel sustains): with 60-bit primes the arithmetic alone caps the algorithmic rate at "
                          "%.0f %% of the HBM peak" % (valu_floor, two_pass, COPY_RATE_GBS, 100 * 16.0 * coeffs / (valu_floor * 1e-3) / 1e9 / HBM_PEAK_GBS))
        out["frac_of_ceiling"] = round(max(two_pass, valu_floor) / fwd_ms, 4)
    return out


def replica_layer(args, env):
    """This rank's replica: one whole encoder layer in a child process bound to the rank's GPU by `env`; (layer_s, None) or (None, why)."""
    text, why = run_child(os.path.join(ROOT, "tools", "cpp", "bench_encoder_layer"), [], args.layer_timeout, "bench_encoder_layer", env)
    line = [l for l in (text or "").splitlines() if l.startswith("LAYER_JSON ")]
    if why or not line:
        return None, why or "bench_encoder_layer printed no result line"
    layer = json.loads(line[-1][len("LAYER_JSON "):])
    if not layer.get("complete"):
        return None, "bench_encoder_layer did not run the whole layer"
    return layer["layer_s"], None


def run_child(exe, argv, timeout, tag, env=None):
    """One child process (never a re-exec of this one); returns (stdout, None) or (None, reason)."""
    import subprocess

    if not os.path.exists(exe):
        return None, "%s is not built on this machine" % os.path.relpath(exe, ROOT)
    try:
        r = subprocess.run([exe] + [str(a) for a in argv], cwd=os.path.dirname(exe), capture_output=True, text=True, timeout=timeout, env=env)
    except subprocess.TimeoutExpired:
        return None, "%s exceeded %.0f s" % (tag, timeout)
    if r.returncode != 0:
        return None, "%s failed (exit %d): %s" % (tag, r.returncode, (r.stderr or r.stdout)[-300:])
    return r.stdout + "\n" + r.stderr, None


def end_to_end_slice(args, cores):
    """The metric's first half, ms per encrypted input, from two child processes at MOAI's parameters (both built by
    __graft_entry__.build(); they include MOAI's own headers, so they are built where the reference checkout exists and travel
    prebuilt):
      tools/cpp/bench_e2e            bootstrap_3 packed and through MOAI's one-call-per-ciphertext pattern; one attention head
                                     and a slice of the feed-forward half through MOAI's UNCHANGED headers;
      tools/cpp/bench_encoder_layer  one WHOLE encoder layer (12 heads, the three products, GELU, two LayerNorms, four
                                     bootstrapping rounds of 768) on 768 ciphertexts = 256 packed inputs, wall time.
    Twelve layers are twelve times one layer (same shapes, same levels: test_full_scheme.hpp:484-1095), the only arithmetic in
    the three headline numbers; the operations bench_e2e counted are priced on the CPU oracle of the same host."""
    t0 = time.perf_counter()
    out = {}
    text, why = run_child(os.path.join(ROOT, "tools", "cpp", "bench_e2e"), [args.e2e_pack, host_cores()], args.e2e_timeout, "bench_e2e")
    line = [l for l in (text or "").splitlines() if l.startswith("E2E_JSON ")]
    if why or not line:
        return {"skipped": why or "bench_e2e printed no result line"}
    child = json.loads(line[-1][len("E2E_JSON "):])
    notes = [l.strip()[:300] for l in text.splitlines()
             if l.startswith(("setup", "bootstrap_3:", "  [device memory", "single_att_block", "Compute Q, K, V", "768 input", "feed-forward slice"))]
    ops_boot = child.pop("ops_bootstrap_pack")
    ops_head = child.pop("ops_head")
    ffn = child.get("ffn_slice") or {}
    out = {
        "workload": "MOAI parameters (N=65536, 36-prime chain {51,46x20,51x14,58}, logn=15, K=25, degree-59 cosine), 768 ciphertexts = 256 "
                    "packed inputs x 128 tokens, synthetic weights: bootstrap_3 on a pack of %d; one attention head and a 128-column slice "
                    "of the feed-forward products + 128 gelu_v2 through MOAI's own headers, unchanged; one whole encoder layer" % child["pack"],
        "bootstrap_ms": child["bootstrap_ms_packed"],
        "bootstrap_ms_moai_call_pattern": child["bootstrap_ms_moai_calls"],
        "bootstrap_max_error": child["bootstrap_max_error"],
        "bootstrap_chain_index_after": child["bootstrap_chain_index_after"],
        "head_s": child["head_s"],
        "head_second_s": child.get("head_second_s"),
        "head_max_error": child["head_max_error"],
        "head_max_error_vs_exact_softmax": child["head_max_error_vs_exact_softmax"],
        "child_wall_s": round(time.perf_counter() - t0, 1),
        "child_setup_s": child["setup_s"],
        "child_log": notes,
    }
    if ffn and ffn.get("selfout_s", -1) > 0:
        # every routine's loop is 128 x (columns / 128) independent columns (Ct_pt_matrix_mul.hpp:60-63, 113-116), GELU is one
        # call per ciphertext: the slice scales by the column count
        c, g = ffn["columns"], ffn["gelu_ciphertexts"]
        out["unchanged_ffn_per_layer_s"] = {
            "selfout": round(ffn["selfout_s"] * 768 / c, 1), "intermediate": round(ffn["intermediate_s"] * 3072 / c, 1),
            "gelu": round(ffn["gelu_s"] * 3072 / g, 1), "final": round(ffn["final_s"] * 768 / c, 1),
            "measured": "a slice of %d columns per product and %d gelu_v2 calls, scaled by the column / ciphertext count" % (c, g),
            "slice_s": {k: ffn[k] for k in ("selfout_s", "intermediate_s", "gelu_s", "final_s")},
        }
    # ---- one whole encoder layer, measured in this run ------------------------------------------------------------------------
    layer = None
    if not args.no_layer:
        t1 = time.perf_counter()
        text, why = run_child(os.path.join(ROOT, "tools", "cpp", "bench_encoder_layer"), [], args.layer_timeout, "bench_encoder_layer")
        line = [l for l in (text or "").splitlines() if l.startswith("LAYER_JSON ")]
        if why or not line:
            out["layer"] = {"skipped": why or "bench_encoder_layer printed no result line"}
        else:
            layer = json.loads(line[-1][len("LAYER_JSON "):])
            layer["child_wall_s"] = round(time.perf_counter() - t1, 1)
            layer["log"] = [l.strip()[:200] for l in text.splitlines() if l.startswith(("keys", "attention", "one encoder layer", "layer output"))]
            out["layer"] = layer
    if layer and layer.get("complete"):
        per_input = lambda layer_s: round(12 * layer_s / 256 * 1e3, 1)
        out["layer_s"] = layer["layer_s"]
        # twelve heads of a layer through MOAI's unchanged header: the first one as measured (it also pays the process's first
        # allocations and hoisting corrections), the other eleven like the second head measured on the same inputs
        h2 = out.get("head_second_s")
        heads_unchanged = out["head_s"] + 11 * h2 if h2 and h2 > 0 else 12 * out["head_s"]
        out["ms_per_input"] = {
            "fused_callers": {
                "value": per_input(layer["layer_s"]),
                "what": "12 x the whole layer measured in this run (tools/cpp/bench_encoder_layer: moai_fused:: products, MOAI's softmax_boot / "
                        "layernorm / gelu_v2 headers on packed ciphertexts, packed bootstrapping) / 256 inputs"},
            "unchanged_attention_fused_ffn": {
                "value": per_input(layer["layer_s"] - layer["attention_s"] + heads_unchanged),
                "what": "the same layer with its attention (%.1f s) replaced by twelve heads through MOAI's unchanged single_att_block.hpp: "
                        "%.2f s for the first + 11 x %.2f s (a second head measured on the same inputs)"
                        % (layer["attention_s"], out["head_s"], h2 if h2 and h2 > 0 else out["head_s"])},
        }
        u = out.get("unchanged_ffn_per_layer_s")
        if u:
            fused_ffn = layer["selfout_s"] + layer["intermediate_s"] + layer["gelu_s"] + layer["final_s"]
            unchanged_ffn = u["selfout"] + u["intermediate"] + u["gelu"] + u["final"]
            boot_calls = 3072 * (out["bootstrap_ms_moai_call_pattern"] - out["bootstrap_ms"]) * 1e-3
            all_unchanged = layer["layer_s"] - layer["attention_s"] + heads_unchanged - fused_ffn + unchanged_ffn + boot_calls
            out["ms_per_input"]["all_callers_unchanged"] = {
                "value": per_input(all_unchanged),
                "what": "what MOAI's all_layer_test (test_full_scheme.hpp:339-1123) would cost unchanged: additionally the four feed-forward "
                        "stages through MOAI's own loops (%.0f s per layer from the scaled slice instead of %.1f s fused) and bootstrap_3 "
                        "called one ciphertext at a time (gathered by the drop-in)" % (unchanged_ffn, fused_ffn)}
        out["paper_ms_per_input_56_cores"] = 574600.0
    if cores:
        try:
            out["cpu_baseline"] = price_on_cpu(ops_boot, ops_head, child["pack"], cores)
            cb = out["cpu_baseline"]
            measured_s = 12 * out["head_s"] + 3072 * out["bootstrap_ms"] * 1e-3
            cpu_measured_s = 12 * cb["head_s"] + 3072 * cb["bootstrap_ms"] * 1e-3
            cb["layer_part_s"] = {"cpu_model": round(cpu_measured_s, 1), "gpu_measured": round(measured_s, 2),
                                  "what": "12 unchanged heads + 3072 bootstraps of one layer; the CPU side is a priced MODEL (operation "
                                          "counts of the GPU run x timed oracle primitives / cores), not a timed CPU run of the pipeline"}
            out["model_ratio_cpu_over_gpu_heads_and_bootstraps"] = round(cpu_measured_s / measured_s, 1)
        except Exception as e:  # the checker must never sink the bench line
            out["cpu_baseline"] = {"skipped": repr(e)[:200]}
    return out


FP64_BFLY_PER_S = 4.5e12  # exact FP64 butterflies per second, whole chip, no memory access (profiles/r01_valu_bench.txt)


def keyswitch_region(m, args, torch, np, dev, stream, cores):
    """BASELINE configs[2]: Evaluator::rotate_vector's key switch (SEAL/evaluator.cpp:2563-2665 over :2724-3020) at N = 2^16 on
    MOAI's chain, batch 256 ciphertexts resident in HBM, one uniform 1.32 GB Galois key, at l = 35 (fresh ciphertexts) and
    l = 15 (where the attention block rotates).  HIP events on the launch stream; ms per ciphertext; against
      * its algorithmic bytes: 5 l N 8 per ciphertext + the key's 2 l (l+1) N 8 once per batch, and
      * its arithmetic: (l^2 + 3l + 2) transforms of N/2 log2 N butterflies + 2 l (l+1) N multiply-accumulates per ciphertext,
        priced at the chip's exact-FP64 butterfly rate (the 46/51-bit primes run on the FP64 pipe; the special prime's rows on the
        integer pipe are 1/(l+1) of the work).
    `traffic` comes from the committed PMC passes over this same region (profiles/), stamped with the commit they were taken at.
    The CPU baseline is the oracle's apply_galois TIMED on this host's cores (one ciphertext per thread), which also checks
    ciphertext 0 of the device result bit for bit.  dnum = 3 of BASELINE's wording has no counterpart in the reference (one digit
    per prime, one special prime): parity is only defined for the reference's decomposition, which this is."""
    primes = moai_primes()
    k = len(primes)
    ctx = m.Context(LOGN, primes, device=dev.index or 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    key = torch.empty((k - 1, 2, k, N), dtype=torch.int64, device=dev)
    for i, q in enumerate(primes):
        key[:, :, i, :] = torch.randint(0, q, (k - 1, 2, N), dtype=torch.int64, device=dev, generator=gen)
    elt = ctx.galois_elt_from_step(1)
    B = args.batch
    res = {"workload": "configs[2]: rotate_vector's key switch (apply_galois, step 1), N=65536, MOAI chain, batch %d ciphertexts, "
                       "one uniform 1.32 GB key; dnum=3 has no counterpart in the reference (parity unpinned by definition), "
                       "the reference's per-prime decomposition is what runs" % B,
           "levels": {}}
    pmc = load_json(os.path.join(ROOT, "profiles", "r03_ks_traffic.json"))
    for Lk in (35, 15):
        ct = torch.empty((B, 2, Lk, N), dtype=torch.int64, device=dev)
        for i in range(Lk):
            ct[:, :, i, :] = torch.randint(0, primes[i], (B, 2, N), dtype=torch.int64, device=dev, generator=gen)
        ct0 = ct[0].clone()
        torch.cuda.synchronize()
        probe = ct0.clone()
        ctx.apply_galois(probe.data_ptr(), Lk, elt, key.data_ptr(), 1, stream=stream)
        ctx.apply_galois(ct.data_ptr(), Lk, elt, key.data_ptr(), B, stream=stream)  # warm-up: grows the workspace arena
        torch.cuda.synchronize()
        ev = [m.hip.Event() for _ in range(2)]
        reps, ms = 3, []
        for _ in range(reps):
            ev[0].record(stream)
            ctx.apply_galois(ct.data_ptr(), Lk, elt, key.data_ptr(), B, stream=stream)
            ev[1].record(stream)
            torch.cuda.synchronize()
            ms.append(ev[1].elapsed_ms_since(ev[0]))
        t = sum(ms) / reps
        alg_bytes = B * 5 * Lk * N * 8 + 2 * Lk * (Lk + 1) * N * 8
        work = B * ((Lk * Lk + 3 * Lk + 2) * (N // 2) * LOGN + 2 * Lk * (Lk + 1) * N)
        lv = {
            "ms_per_ciphertext": round(t / B, 4),
            "ms_per_batch": round(t, 2),
            "algorithmic_bytes_per_launch": alg_bytes,
            "roofline": {
                "bound": "valu (FP64 pipe as an exact integer unit); not HBM: l^2 transforms per ciphertext",
                "achieved": round(work / (t * 1e-3) / 1e9, 1), "peak": FP64_BFLY_PER_S / 1e9, "unit": "G(butterfly+MAC)/s",
                "frac": round(work / (t * 1e-3) / FP64_BFLY_PER_S, 4),
                "hbm_algorithmic_gbs": round(alg_bytes / (t * 1e-3) / 1e9, 1),
                # HBM bytes of one batch call from the committed PMC passes (per ciphertext at batch 64, times this batch; the key's
                # share is amortised differently at other batch sizes -- a few per cent); null where no PMC run exists
                "traffic": ((pmc or {}).get("l%d" % Lk) or {}).get("hbm_bytes_per_ciphertext", 0) * B or None,
                "traffic_source": ("profiles/r03_ks_traffic.json, collected at commit %s" % (pmc or {}).get("collected_at_commit")) if (pmc or {}).get("l%d" % Lk) else None,
            },
        }
        if cores:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle as O  # checker + baseline only
            from concurrent.futures import ThreadPoolExecutor

            octx = O.Context(LOGN, primes)
            O.lib().mo_set_threads(1)
            hkey = key.cpu().numpy().view(np.uint64)
            hct = ct0.cpu().numpy().view(np.uint64)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                outs = list(ex.map(lambda i: octx.apply_galois(hct, Lk, elt, hkey), range(cores)))
            el = time.perf_counter() - t0
            got = probe.cpu().numpy().view(np.uint64)
            assert (got == outs[0].reshape(2, Lk, N)).all(), "GPU key switch differs from the oracle at l = %d" % Lk
            lv["cpu_baseline"] = {"value": round(el / cores * 1e3, 2), "unit": "ms per ciphertext (host throughput, all cores busy)",
                                  "cores": cores, "kind": "port",
                                  "sample": "%d ciphertexts, one per thread, oracle apply_galois at l=%d: %.1f s wall" % (cores, Lk, el),
                                  "parity": "ciphertext 0 of the device result equals the oracle's, bit for bit"}
        res["levels"]["l%d" % Lk] = lv
        del ct
    ctx.close()
    del key
    torch.cuda.empty_cache()
    return res


def ntt_moai_chain_region(m, args, torch, np, dev, stream, check):
    """The same microbenchmark on the chain MOAI really runs (test_full_scheme.hpp:356-378): N = 2^16, the 35 data primes
    {51, 46 x 20, 51 x 14}, batch ciphertexts x 2 polynomials.  Every one of these primes is below 2^51, so the transforms run
    their butterflies on the FP64 pipe as an exact integer unit (csrc/modarith.hip.h M_FPN / M_FPR): 4.5e12 butterflies/s
    instead of 1.7e12, and what binds is the HBM traffic of the two launches -- the HBM fraction of the REAL workload."""
    primes = moai_primes()[:-1]
    Lm = len(primes)
    ctx = m.Context(LOGN, moai_primes(), device=dev.index or 0)
    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    data = torch.empty((B, 2, Lm, N), dtype=torch.int64, device=dev)
    for i, q in enumerate(primes):
        data[:, :, i, :] = torch.randint(0, q, (B, 2, N), dtype=torch.int64, device=dev, generator=gen)
    before = data[0, 0].clone()
    ptr, n_poly = data.data_ptr(), B * 2
    ctx.ntt_forward(ptr, n_poly, Lm, stream=stream)
    ctx.ntt_inverse(ptr, n_poly, Lm, stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(data[0, 0], before), "INTT(NTT(x)) != x on MOAI's chain"
    ev = [m.hip.Event() for _ in range(3)]
    fwd, inv, reps = 0.0, 0.0, 5
    for _ in range(reps):
        ev[0].record(stream)
        ctx.ntt_forward(ptr, n_poly, Lm, stream=stream)
        ev[1].record(stream)
        ctx.ntt_inverse(ptr, n_poly, Lm, stream=stream)
        ev[2].record(stream)
        torch.cuda.synchronize()
        fwd += ev[1].elapsed_ms_since(ev[0]) / reps
        inv += ev[2].elapsed_ms_since(ev[1]) / reps
    if check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle as O  # checker only

        x = before.clone()
        ctx.ntt_forward(x.data_ptr(), 1, Lm, stream=stream)
        torch.cuda.synchronize()
        want = O.Context(LOGN, moai_primes()).ntt(before.cpu().numpy().view(np.uint64).reshape(1, Lm, N), Lm)[0]
        assert (x.cpu().numpy().view(np.uint64) == want).all(), "GPU forward NTT differs from the oracle on MOAI's chain"
    coeffs = n_poly * Lm * N
    bytes_dir = 16.0 * coeffs
    bfly = coeffs / 2 * LOGN
    traffic = (load_json(os.path.join(ROOT, "profiles", "pmc_traffic.json")) or {}).get("ntt_moai_chain_forward_hbm_bytes_per_launch")
    out = {
        "workload": "negacyclic NTT then INTT, N=65536, MOAI's 35 data primes {51, 46x20, 51x14}, %d ciphertexts x 2 polys, FP64-pipe butterflies" % B,
        "value": round(2 * bytes_dir / ((fwd + inv) * 1e-3) / 1e9, 1), "unit": "GB/s",
        "roofline": {
            "bound": "hbm", "kernel": "forward NTT = ntt_fwd_strided<16,2|3> + ntt_fwd_contig<16,2|3>; inverse = ntt_inv_contig<16,2|3> + ntt_inv_strided<16,2|3> (exact FP64 butterflies both ways)",
            "achieved": round(bytes_dir / (fwd * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(bytes_dir / (fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic,
            "fwd_ms": round(fwd, 4), "inv_ms": round(inv, 4), "inv_achieved": round(bytes_dir / (inv * 1e-3) / 1e9, 1),
            "algorithmic_bytes_per_launch": bytes_dir,
            "floors_ms": {"hbm_algorithmic_at_spec_peak": round(bytes_dir / HBM_PEAK_GBS / 1e6, 3),
                          "valu_fp64": round(bfly / FP64_BFLY_PER_S * 1e3, 3),
                          "two_pass_traffic_at_copy_rate": round(2 * bytes_dir / COPY_RATE_GBS / 1e6, 3)},
        },
    }
    ctx.close()
    del data
    torch.cuda.empty_cache()
    return out


def load_json(path):
    try:
        with open(path) as f:
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

    unpriced = set()

    def price(ops):
        total, by = 0.0, {}
        for name, L, units in ops:
            if name in ("apply_galois_to", "apply_galois_hoisted", "switch_key", "relinearize"):
                # apply_galois (in place) and apply_galois_acc are counted as the apply_galois_to they call; a hoisted call counts
                # batch x rotations: the reference makes one full key switch per rotation
                c = units * t_ks(L)
            elif name == "rescale":
                c = units / 2 * t_lin("rescale", L)
            elif name == "rescale_add":
                c = units / 2 * (t_lin("rescale", L) + t_lin("add", L))
            elif name == "mul_scalar_rescale":
                c = units / 2 * (t_lin("rescale", L) + t_lin("mulplain", L))
            elif name == "mul_scalar_rescale_add":
                c = units / 2 * (t_lin("rescale", L) + t_lin("mulplain", L) + t_lin("add", L))
            elif name in ("ct_multiply", "ct_square", "ct_multiply_general"):
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
            elif name in ("ckks_encode", "ckks_encode_masked", "hoist_correction"):
                # the reference encodes a plaintext per product (an FP64 transform + L NTTs each); not priced, which favours the
                # CPU.  hoist_correction is a per-key constant of this library with no counterpart in the reference.
                c = 0.0
            else:
                unpriced.add(name)
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
        "operations_without_a_price": sorted(unpriced),
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
