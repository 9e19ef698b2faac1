"""CPU, world_size 2 over gloo: the N>1 path of bench.py (tuple sharding + the one metric collective)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tuples, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from e2e_multi_view_matching_amd.distributed import gather_pair_errors, reduce_max_seconds, shard_range
    from e2e_multi_view_matching_amd.metrics import pose_auc
    lo, hi = shard_range(n_tuples, rank, world)
    all_err = np.random.default_rng(0).uniform(0, 30, n_tuples)
    all_err[3] = np.inf  # a pair whose pose could not be computed
    gathered = gather_pair_errors(all_err[lo:hi])
    t = reduce_max_seconds(1.0 + rank)
    dist.barrier()
    q.put((rank, gathered.tolist(), pose_auc(gathered, [5, 10, 20]), t, (lo, hi)))
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process_auc():
    from e2e_multi_view_matching_amd.metrics import pose_auc
    world, n = 2, 37  # uneven shards: 19 + 18
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = np.random.default_rng(0).uniform(0, 30, n)
    ref[3] = np.inf
    for rank, gathered, auc, t, span in res:
        assert np.allclose(np.array(gathered, dtype=np.float32), ref.astype(np.float32))
        assert np.allclose(auc, pose_auc(ref.astype(np.float32), [5, 10, 20]))
        assert t == 2.0  # MAX over ranks
    assert res[0][4] == (0, 19) and res[1][4] == (19, 37)
