"""Multi-GPU sharding of the hot path: independent ciphertext batches per rank, no data-path collective.

The path partitions on the ciphertext index: every (ciphertext, polynomial, prime) row is
independent, keys and tables are replicated read-only.  torch.distributed (RCCL on the GPU box, gloo in
the CPU tests) is used for exactly two things: a barrier around the timed region and the max-over-ranks
of the elapsed time."""


def shard_range(total, rank, world):
    """Contiguous [start, start+count) of `total` ciphertexts for `rank`; remainders go to the low ranks."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def barrier(dist):
    if dist is not None and dist.is_initialized():
        dist.barrier()


def max_over_ranks(seconds, dist, device=None):
    """Wall time of the slowest rank (the job is done when the last rank is)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch

    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(units, dist, device=None):
    """Units all ranks processed together."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(units)
    import torch

    u = torch.tensor([float(units)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item())


def whole_job_rate(units_this_rank, seconds_this_rank, dist, device=None):
    """(sum over ranks of units) / (max over ranks of seconds): the aggregate the bench reports."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_this_rank / seconds_this_rank
    import torch

    dev = device if device is not None else "cpu"
    u = torch.tensor([float(units_this_rank)], dtype=torch.float64, device=dev)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / max_over_ranks(seconds_this_rank, dist, device)


# ---- end-to-end replicas ("replicas only") ------------------------------------------------------------
# The 256 inputs of a batch are slot-packed into EVERY ciphertext (include/source/matrix_mul/Batch_encode_encrypt.hpp:21-28),
# so the encrypted forward pass does not shard by input: multi-GPU = one independent packed batch (256 inputs, its own keys)
# per GPU, no exchange step at all.  bench.py starts one child process per rank, bound to that rank's device BEFORE the child
# makes any GPU call, and sums the replicas' rates.

def replica_env(local_rank, environ):
    """Environment of the rank's child process: the child must see exactly the parent rank's GPU as its device 0.  If the launcher
    already restricted the parent (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES), local_rank indexes that list."""
    env = dict(environ)
    chosen = str(int(local_rank))
    for name in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = [d for d in env.get(name, "").split(",") if d != ""]
        if listed:
            if not (0 <= int(local_rank) < len(listed)):
                raise ValueError("%s=%s has no entry for local rank %d" % (name, env[name], local_rank))
            chosen = listed[int(local_rank)]
            break
    env["HIP_VISIBLE_DEVICES"] = chosen
    env.pop("CUDA_VISIBLE_DEVICES", None)  # the HIP runtime honours both; leave one statement of the binding
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def aggregate_replicas(layer_s, inputs_per_replica, layers, dist, device=None):
    """Whole-job figures of N independent replicas that each measured `layer_s` seconds per encoder layer (None = this rank's
    replica failed): inputs per second summed over the replicas that ran, the slowest and fastest layer, and how many ran.
    ms per input of the job = 1000 / (summed inputs per second)."""
    ok = layer_s is not None and layer_s > 0
    rate = inputs_per_replica / (layers * layer_s) if ok else 0.0
    total_rate = sum_over_ranks(rate, dist, device)
    ran = int(round(sum_over_ranks(1.0 if ok else 0.0, dist, device)))
    slowest = max_over_ranks(layer_s if ok else 0.0, dist, device)
    fastest = -max_over_ranks(-(layer_s if ok else 1e30), dist, device)
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    return {
        "n_gpus": world,
        "replicas_completed": ran,
        "inputs_per_s": total_rate,
        "ms_per_input": (1e3 / total_rate) if total_rate > 0 else None,
        "layer_s_slowest": slowest if ran else None,
        "layer_s_fastest": fastest if ran else None,
    }
