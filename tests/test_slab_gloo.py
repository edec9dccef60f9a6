"""Multi-process (gloo, CPU) tests of the z-slab halo-exchange logic of tomobar_amd.slab: every rank owns a
slab of a seeded volume, runs the slab TV drivers with the ORACLE's single-iteration functions as the compute step, and
checks its slab against the oracle's whole-volume result -- bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import tomo_oracle as O
        from tomobar_amd.slab import SlabComm, pd_tv_slab, rof_tv_slab, slab_bounds
        comm = SlabComm(rank, world)
        nz, dy, dx = case["shape"]
        rng = np.random.default_rng(5)
        vol = (rng.random((nz, dy, dx)) * 0.3 + (np.indices((nz, dy, dx))[2] > dx // 2)).astype(np.float32)
        z0, z1 = slab_bounds(nz, world, rank)
        mine = torch.from_numpy(vol[z0:z1].copy())
        if case["kind"] == "pd":
            want = O.pd_tv(vol, 0.04, case["iters"], case["mtv"], case["nn"], 8.0, case["half"])
            got = pd_tv_slab(mine, comm, 0.04, case["iters"], case["mtv"], case["nn"], 8.0, case["half"],
                             pair_fn=O.pd_pair_slab, step_fn=O.pd_step_slab)
        else:
            want = O.rof_tv(vol, 0.05, case["iters"], 0.005, case["half"])
            got = rof_tv_slab(mine, comm, 0.05, case["iters"], 0.005, case["half"], step_fn=O.rof_step_slab)
        assert np.array_equal(got.numpy(), want[z0:z1]), (rank, np.abs(got.numpy() - want[z0:z1]).max())
        # scalar reductions used by the power method / PWLS / CGLS
        assert comm.allreduce_sum(float(rank + 1)) == world * (world + 1) / 2
        assert comm.allreduce_max(float(rank)) == world - 1
    finally:
        dist.destroy_process_group()


CASES = [
    dict(kind="pd", shape=(9, 7, 11), iters=6, mtv=0, nn=0, half=False),
    dict(kind="pd", shape=(8, 6, 70), iters=5, mtv=1, nn=1, half=True),
    # slabs long enough for the overlapped schedule (edge planes, exchange in flight, interior); with 3 ranks the
    # middle slab is too short and falls back to the plain schedule while its neighbours overlap
    dict(kind="pd", shape=(19, 6, 13), iters=7, mtv=0, nn=1, half=False),
    dict(kind="pd", shape=(26, 5, 12), iters=6, mtv=0, nn=0, half=True),
    dict(kind="rof", shape=(9, 7, 11), iters=6, half=False),
    dict(kind="rof", shape=(22, 6, 10), iters=5, half=False),   # long enough for the overlapped schedule
    dict(kind="rof", shape=(10, 5, 9), iters=4, half=True),
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['kind']}-{'x'.join(map(str, c['shape']))}-h{int(c['half'])}")
def test_slab_tv_matches_whole_volume(world, case):
    mp.start_processes(_worker, args=(world, _free_port(), case), nprocs=world, join=True, start_method="spawn")


def test_slab_bounds_cover_volume():
    from tomobar_amd.slab import slab_bounds
    for nz, world in ((10, 3), (1024, 8), (2160, 8), (5, 8)):
        b = [slab_bounds(nz, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == nz
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
