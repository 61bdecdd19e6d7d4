"""The N > 1 path of bench.py on CPU: world_size 2 over gloo.  The data path has no collective; what is
distributed is the sharding of the ciphertext batch and the barrier / max-over-ranks timing."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import __graft_entry__ as g

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = g.load_package()
    start, count = m.shard.shard_range(257, rank, world)
    m.shard.barrier(dist)
    # rank 1 is "slower": the job takes as long as the slowest rank
    seconds = 2.0 if rank == 1 else 1.0
    t = m.shard.max_over_ranks(seconds, dist)
    rate = m.shard.whole_job_rate(float(count), seconds, dist)
    # what `bench.py --gpus 2 --split` transforms on this rank (strong scaling: one 256-ciphertext batch shared), and
    # what it transforms without --split (weak scaling: a whole batch per rank); the job's rows are summed over ranks
    import bench

    split = bench.batch_of_rank(256, rank, world, True, m.shard)
    weak = bench.batch_of_rank(256, rank, world, False, m.shard)
    job_split = m.shard.sum_over_ranks(float(split[0]), dist)
    job_weak = m.shard.sum_over_ranks(float(weak[0]), dist)
    # the end-to-end replicas of `bench.py --gpus 2`: each rank measured its own encoder layer (rank 1's GPU was slower); the
    # job's inputs per second are summed, no exchange step
    agg = m.shard.aggregate_replicas(200.0 if rank == 0 else 250.0, 256, 12, dist)
    # ... and when one replica fails the others still count
    agg_partial = m.shard.aggregate_replicas(None if rank == 1 else 200.0, 256, 12, dist)
    m.shard.barrier(dist)
    q.put((rank, start, count, t, rate, split, weak, job_split, job_weak, agg, agg_partial))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, c0, t0, rate0, split0, weak0, js0, jw0, agg0, part0), (r1, s1, c1, t1, rate1, split1, weak1, js1, jw1, agg1, part1) = res
    # replicas: 256 / (12 x 200) + 256 / (12 x 250) inputs per second, the same figure on both ranks
    assert agg0 == agg1 and agg0["n_gpus"] == 2 and agg0["replicas_completed"] == 2
    assert agg0["inputs_per_s"] == pytest.approx(256 / 2400.0 + 256 / 3000.0)
    assert agg0["ms_per_input"] == pytest.approx(1e3 / (256 / 2400.0 + 256 / 3000.0))
    assert agg0["layer_s_slowest"] == 250.0 and agg0["layer_s_fastest"] == 200.0
    assert part0 == part1 and part0["replicas_completed"] == 1 and part0["inputs_per_s"] == pytest.approx(256 / 2400.0)
    assert part0["layer_s_slowest"] == part0["layer_s_fastest"] == 200.0
    # --split: rank 0 transforms ciphertexts [0, 128), rank 1 [128, 256) of the ONE batch; without it 256 each
    assert split0 == (128, 0) and split1 == (128, 128)
    assert weak0 == (256, 0) and weak1 == (256, 256)
    assert js0 == js1 == 256.0 and jw0 == jw1 == 512.0
    assert (s0, c0) == (0, 129) and (s1, c1) == (129, 128)  # disjoint, covers all 257 ciphertexts
    assert t0 == t1 == 2.0  # max over ranks
    assert rate0 == rate1 == pytest.approx(257 / 2.0)  # all units / slowest rank


def test_shard_range_properties():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    m = g.load_package()
    for total in (0, 1, 7, 256, 257, 768):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                s, c = m.shard.shard_range(total, r, world)
                cover += list(range(s, s + c))
            assert cover == list(range(total))
    with pytest.raises(ValueError):
        m.shard.shard_range(8, 2, 2)


def test_replica_child_is_bound_to_the_ranks_device():
    """bench.py --gpus N starts one end-to-end child per rank; the child must see the rank's GPU as its only device, whatever
    the launcher already restricted."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    m = g.load_package()
    env = m.shard.replica_env(3, {"PATH": "/bin"})
    assert env["HIP_VISIBLE_DEVICES"] == "3" and env["PATH"] == "/bin" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    env = m.shard.replica_env(1, {"HIP_VISIBLE_DEVICES": "4,6,7"})
    assert env["HIP_VISIBLE_DEVICES"] == "6"
    env = m.shard.replica_env(0, {"CUDA_VISIBLE_DEVICES": "5", "HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert env["HIP_VISIBLE_DEVICES"] == "5" and "CUDA_VISIBLE_DEVICES" not in env and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    with pytest.raises(ValueError):
        m.shard.replica_env(2, {"HIP_VISIBLE_DEVICES": "0,1"})
    # one replica, no process group: the rank's own figures
    agg = m.shard.aggregate_replicas(192.0, 256, 12, None)
    assert agg["n_gpus"] == 1 and agg["replicas_completed"] == 1 and agg["ms_per_input"] == pytest.approx(12 * 192.0 / 256 * 1e3)
