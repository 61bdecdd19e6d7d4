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
