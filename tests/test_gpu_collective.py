"""The library-owned metric collective (csrc/comm.hip: e2emv_comm_* / e2emv_metric_allgather / e2emv_metric_allreduce over RCCL) -
the reference's init_process_group(backend="nccl") + all_reduce (train.py:270-277, 102-106) for a host without torch.distributed.
One rank per GPU is RCCL's rule, so on the 1-GPU box the communicator has one rank (both bootstrap forms); with two or more GPUs two
processes gather through a file-bootstrapped communicator.  The torch.distributed path of the same gather runs over gloo in
tests/test_distributed_gloo.py."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_communicator_both_bootstraps(gpu):
    from e2e_multi_view_matching_amd.distributed import LibraryComm
    uid = LibraryComm.unique_id(gpu)
    assert len(uid) == 128 and any(uid)
    c = LibraryComm(0, 1, gpu, id_bytes=uid)
    x = torch.arange(37, dtype=torch.float32, device=gpu) * 0.5
    g = c.allgather(x)
    assert g.shape == (1, 37) and torch.equal(g[0], x)
    y = x.clone()
    for op in ("sum", "max", "min"):
        assert torch.equal(c.allreduce_(y, op), x)
    c.close()
    with tempfile.TemporaryDirectory() as d:
        c2 = LibraryComm(0, 1, gpu, id_file=os.path.join(d, "id"))
        assert os.path.getsize(os.path.join(d, "id")) == 128
        assert torch.equal(c2.allgather(x)[0], x)
        c2.close()


def test_bad_arguments_are_errors_not_hangs(gpu):
    from e2e_multi_view_matching_amd import _lib
    from e2e_multi_view_matching_amd.distributed import LibraryComm
    with pytest.raises(_lib.E2EMVError):
        LibraryComm(3, 2, gpu, id_bytes=bytes(128))  # rank outside the world
    with tempfile.TemporaryDirectory() as d:
        with pytest.raises(_lib.E2EMVError, match="waited"):
            LibraryComm(1, 2, gpu, id_file=os.path.join(d, "never_written"), timeout_s=0.2)  # rank 0 never shows up


_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[4])
from e2e_multi_view_matching_amd.distributed import LibraryComm
rank, world, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dev = torch.device("cuda", rank)
torch.cuda.set_device(dev)
c = LibraryComm(rank, world, dev, id_file=path)
g = c.allgather(torch.full((5,), float(rank + 1), device=dev))
s = c.allreduce_(torch.full((3,), float(rank + 1), device=dev), "sum")
torch.cuda.synchronize()
assert g.shape == (world, 5) and all(float(g[r, 0]) == r + 1 for r in range(world)), g
assert float(s[0]) == world * (world + 1) / 2, s
c.close()
print("ok", rank)
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL wants one GPU per rank: needs two GPUs")
def test_two_ranks_two_gpus_file_bootstrap(gpu):
    with tempfile.TemporaryDirectory() as d:
        ps = [subprocess.Popen([sys.executable, "-c", _WORKER, str(r), "2", os.path.join(d, "id"), ROOT], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=300)[0] for p in ps]
        assert all(p.returncode == 0 for p in ps), outs
