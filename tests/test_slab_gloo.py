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


def _seams():
    """CPU stand-ins for the C-ABI calls of tomobar_amd.slab (halo pack / unpack, iterations per launch)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _cpu_backend
    _cpu_backend.install()


def _worker(rank, world, port, case):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _seams()
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
        # one message each way per neighbour and exchange, whatever the number of arrays (U, P1, P2, P3) it carries
        st = comm.timing_summary()
        assert st["messages"] == 2 * st["exchanges"] * (int(comm.has_lo) + int(comm.has_hi)) or world == 1, st
        # scalar reductions used by the power method / PWLS / CGLS
        assert comm.allreduce_sum(float(rank + 1)) == world * (world + 1) / 2
        assert comm.allreduce_max(float(rank)) == world - 1
    finally:
        dist.destroy_process_group()


CASES = [
    dict(kind="pd", shape=(9, 7, 11), iters=6, mtv=0, nn=0, half=False),
    dict(kind="pd", shape=(10, 6, 70), iters=5, mtv=1, nn=1, half=True),
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


def _inflight_worker(rank, world, port):
    """exchange_start / exchange_wait allow several exchanges in flight; each must own its staging buffers
    from post to wait (a shared per-direction buffer would be overwritten by the second post).  Host tensors: packed by
    slab.py itself, without the library."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tomobar_amd.slab import SlabComm
        comm = SlabComm(rank, world)
        lo, hi = rank > 0, rank < world - 1

        def blocks(tag, sizes):
            return [torch.full((k, 3, 5), float(100 * tag + 10 * rank + i)) for i, k in enumerate(sizes)]
        sets = []
        for tag, (up, down) in enumerate([((3, 3, 3, 3), (3, 2, 2, 2)), ((1, 4), (2,)), ((3, 3, 3, 3), (3, 2, 2, 2))]):
            su = blocks(tag, up) if hi else []
            sd = blocks(tag, down) if lo else []
            ru = [torch.zeros((k, 3, 5)) for k in down] if hi else []
            rd = [torch.zeros((k, 3, 5)) for k in up] if lo else []
            sets.append((tag, sd, rd, su, ru, up, down))
        handles = [comm.exchange_start(sd, rd, su, ru) for _, sd, rd, su, ru, _, _ in sets]   # three posts, nothing waited yet
        for h in reversed(handles):                                                           # ... completed out of order
            comm.exchange_wait(h)
        for tag, sd, rd, su, ru, up, down in sets:
            for i, t in enumerate(rd):
                assert torch.all(t == float(100 * tag + 10 * (rank - 1) + i)), (rank, tag, i)
            for i, t in enumerate(ru):
                assert torch.all(t == float(100 * tag + 10 * (rank + 1) + i)), (rank, tag, i)
        # the buffers went back to the pool and are reused by the next exchange (no growth beyond what was in flight)
        pooled = len(comm._stage_free)
        comm.exchange(sets[0][1], sets[0][2], sets[0][3], sets[0][4])
        assert len(comm._stage_free) == pooled
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_several_exchanges_in_flight(world):
    mp.start_processes(_inflight_worker, args=(world, _free_port()), nprocs=world, join=True, start_method="spawn")


def _fista_worker(rank, world, port, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _cpu_backend
        from oracle import tomo_oracle as O
        _cpu_backend.install()
        from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
        from tomobar_amd.slab import SlabComm, slab_bounds
        nz, n, na, os_n = case["nz"], 20, 24, case["os"]
        angles = np.linspace(0, np.pi, na, endpoint=False)
        rng = np.random.default_rng(2)
        sino = np.abs(O.shepp_logan_sino(n, nz, n, angles) / n + 0.02 * rng.standard_normal((nz, na, n))).astype(np.float32)
        cor = 0.0
        if case.get("vshift"):   # per-angle (horizontal, vertical) offsets: detector rows are resampled ACROSS the slab boundary
            cor = np.stack([np.linspace(-1.0, 1.5, na), case["vshift"] * np.cos(np.linspace(0.3, 2.9, na))], axis=1)
        P = O.Projector(nz, n, n, angles, cor, os_n)
        z0, z1 = slab_bounds(nz, world, rank)
        rt = RecToolsIRCuPy(n, 0, z1 - z0, cor, angles, n, 0, os_n if os_n > 1 else None)
        rt.slab = SlabComm(rank, world)
        # power method: the eigenvector spans all slabs (norm all-reduced every iteration)
        rt.power_seed = 3
        L_slab = rt.powermethod({"projection_data": None})
        L_whole = O.power_method(P, rng.standard_normal((nz, n, n)).astype(np.float32))
        starts = []   # the start vector the ranks drew, slab by slab: the same iteration on the whole volume
        for r in range(world):
            g = torch.Generator()
            g.manual_seed(3)
            a, b = slab_bounds(nz, world, r)
            starts.append(torch.randn((b - a, n, n), dtype=torch.float32, generator=g).numpy())
        np.testing.assert_allclose(L_slab, O.power_method(P, np.concatenate(starts, axis=0)), rtol=1e-4)
        # (with a vertical component the slices are coupled: 15 iterations from ANOTHER start agree to a few per cent only)
        np.testing.assert_allclose(L_slab, L_whole, rtol=5e-2 if case.get("vshift") else 1e-4)
        reg = dict(case["reg"]) if case["reg"] else None
        full_reg = None if reg is None else {"regul_param": 0.001, "iterations": 150, "time_marching_step": 0.005,
                                             "PD_LipschitzConstant": 12.0, "methodTV": 0, **reg}
        d = {"projection_data": torch.from_numpy(sino[z0:z1].copy()), "data_axes_labels_order": ["detY", "angles", "detX"],
             "data_fidelity": case["fid"]}
        ring = case.get("ring")
        if ring:   # Group-Huber offsets [detY, detX] and their angle sums are slab-local; the PWLS maximum is all-reduced
            d.update(ringGH_lambda=ring["lambda"], ringGH_accelerate=ring["accelerate"])
        if case["fid"] == "SWLS":
            d["beta_SWLS"] = 0.3
        robust = case.get("robust", {})   # Huber / Student's-t re-weighting: element-wise, hence slab-local
        for key, val in robust.items():
            d[f"{key}_threshold"] = val
        if case["method"] == "FISTA":
            want = O.fista(P, sino, 2, L_whole, True, full_reg, case["fid"], ring=ring, beta_swls=0.3, **robust)
            got = rt.FISTA(d, {"iterations": 2, "lipschitz_const": L_whole, "nonnegativity": True,
                               "recon_mask_radius": None}, reg)
        else:
            want = O.admm(P, sino, 3, L_whole, 1.0, 1.6, False, full_reg, case["fid"])
            got = rt.ADMM(d, {"iterations": 3, "lipschitz_const": L_whole, "recon_mask_radius": None}, reg)
        assert np.array_equal(got.numpy(), want[z0:z1]), (rank, float(np.abs(got.numpy() - want[z0:z1]).max()))
    finally:
        dist.destroy_process_group()


FISTA_CASES = [
    dict(method="FISTA", nz=9, os=4, fid="PWLS", reg=dict(method="PD_TV", regul_param=0.002, iterations=7)),
    dict(method="FISTA", nz=8, os=1, fid="LS", reg=dict(method="ROF_TV", regul_param=0.002, iterations=5,
                                                        time_marching_step=0.002)),
    dict(method="ADMM", nz=10, os=3, fid="LS", reg=dict(method="PD_TV", regul_param=0.004, iterations=6)),
    # ring-artefact data terms in z-slab mode (BASELINE configs[4] runs FISTA-OS + PD_TV + the Group-Huber term on 8 ranks)
    dict(method="FISTA", nz=10, os=4, fid="PWLS", ring={"lambda": 1e-4, "accelerate": 3},
         reg=dict(method="PD_TV", regul_param=0.002, iterations=6)),
    dict(method="FISTA", nz=9, os=3, fid="SWLS", reg=None),
    # robust data terms (Huber with the ring term, Student's t plain) in z-slab mode
    dict(method="FISTA", nz=9, os=3, fid="PWLS", ring={"lambda": 1e-4, "accelerate": 3}, robust={"huber": 0.3},
         reg=dict(method="PD_TV", regul_param=0.002, iterations=6)),
    dict(method="FISTA", nz=8, os=2, fid="LS", robust={"studentst": 1.5}, reg=None),
    # ADMM + ROF_TV without subsets: what BASELINE configs[3] runs on 4 ranks
    dict(method="ADMM", nz=9, os=1, fid="LS", reg=dict(method="ROF_TV", regul_param=0.003, iterations=5, time_marching_step=0.002)),
    # vertical CoR component: ghost detector rows travel with every projector call (tomobar_amd.slab.extend_detector_rows)
    dict(method="FISTA", nz=11, os=3, fid="PWLS", vshift=1.7, reg=dict(method="PD_TV", regul_param=0.002, iterations=6)),
    dict(method="ADMM", nz=9, os=1, fid="LS", vshift=0.6, reg=dict(method="ROF_TV", regul_param=0.002, iterations=4,
                                                                    time_marching_step=0.002)),
]


@pytest.mark.parametrize("case", FISTA_CASES, ids=lambda c: f"{c['method']}-os{c['os']}-{c['fid']}-{(c['reg'] or {}).get('method')}"
                         + ("-vertical-cor" if c.get("vshift") else "") + ("-ring" if c.get("ring") else "")
                         + "".join(f"-{k}" for k in c.get("robust", {})))
def test_slab_reconstruction_drivers_match_whole_volume(case):
    """world-2 gloo run of RecToolsIRCuPy.powermethod / FISTA / ADMM with ``rt.slab`` set: power-method all-reduce, PWLS
    maximum all-reduce and the slab proximal step together; the oracle stands in for the C-ABI library at the projector /
    ops seams (tests/_cpu_backend.py), every driver line is the product's."""
    mp.start_processes(_fista_worker, args=(2, _free_port(), case), nprocs=2, join=True, start_method="spawn")


def _short_slab_worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _seams()
        from oracle import tomo_oracle as O
        from tomobar_amd.slab import SlabComm, pd_tv_slab, rof_tv_slab, slab_bounds
        comm = SlabComm(rank, world)
        z0, z1 = slab_bounds(5, world, rank)          # 2 + 2 + 1 slices: too short for the ghost zones
        mine = torch.zeros((z1 - z0, 4, 6))
        for fn, kw in ((pd_tv_slab, dict(pair_fn=O.pd_pair_slab, step_fn=O.pd_step_slab)),
                       (rof_tv_slab, dict(step_fn=O.rof_step_slab))):
            try:
                if fn is pd_tv_slab:
                    fn(mine, comm, 0.04, 4, 0, 0, 8.0, False, **kw)
                else:
                    fn(mine, comm, 0.04, 4, 0.005, False, **kw)
            except ValueError as e:   # EVERY rank raises (nobody is left waiting in a send/recv)
                assert "fewer than" in str(e)
            else:
                raise AssertionError(f"rank {rank}: a too-short slab must raise on all ranks")
        # ghost DETECTOR rows of a vertical CoR component: rank 2 holds one row, its neighbour needs three
        from tomobar_amd.slab import check_ghost_rows, extend_detector_rows
        try:
            check_ghost_rows(comm, 3, z1 - z0, "a vertical CoR component")
        except ValueError as e:
            assert "thinner" in str(e)
        else:
            raise AssertionError(f"rank {rank}: a too-thin slab must raise on all ranks")
        ext, lo = extend_detector_rows(comm, torch.full((z1 - z0, 2, 3), float(rank + 1)), 1)
        assert lo == (1 if rank > 0 else 0) and ext.shape[0] == (z1 - z0) + lo + (1 if rank < world - 1 else 0)
        assert rank == 0 or float(ext[0, 0, 0]) == rank              # the lower neighbour's last row
        assert rank == world - 1 or float(ext[-1, 0, 0]) == rank + 2  # the upper neighbour's first row
        dist.barrier()                                 # all ranks got here: no deadlock
    finally:
        dist.destroy_process_group()


def test_short_slab_raises_on_every_rank_instead_of_deadlocking():
    mp.start_processes(_short_slab_worker, args=(3, _free_port()), nprocs=3, join=True, start_method="spawn")


def test_slab_bounds_cover_volume():
    from tomobar_amd.slab import slab_bounds
    for nz, world in ((10, 3), (1024, 8), (2160, 8), (5, 8)):
        b = [slab_bounds(nz, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == nz
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [z1 - z0 for z0, z1 in b]
        assert max(sizes) - min(sizes) <= 1, "balanced split: no rank is short while another holds a full share"
    from tomobar_amd.slab import check_slab_split
    check_slab_split(16, 8)
    with pytest.raises(ValueError):
        check_slab_split(15, 8)


def _two_volumes_worker(rank, world, port):
    """ONE communicator, two volumes whose balanced splits differ -- 9 slices over 2 ranks is
    5 + 4, 8 slices is 4 + 4.  With a per-rank cache keyed on the local height rank 1 (4, 4) skipped the second collective
    while rank 0 (5, 4) entered it: a hang.  The check is now collective on every call."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _seams()
        from oracle import tomo_oracle as O
        from tomobar_amd.slab import SlabComm, pd_tv_slab, slab_bounds
        comm = SlabComm(rank, world)
        for nz in (9, 8, 9):
            rng = np.random.default_rng(nz)
            vol = rng.random((nz, 6, 9)).astype(np.float32)
            z0, z1 = slab_bounds(nz, world, rank)
            got = pd_tv_slab(torch.from_numpy(vol[z0:z1].copy()), comm, 0.04, 5, 0, 0, 8.0, False,
                             pair_fn=O.pd_pair_slab, step_fn=O.pd_step_slab)
            want = O.pd_tv(vol, 0.04, 5, 0, 0, 8.0, False)
            assert np.array_equal(got.numpy(), want[z0:z1])
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_one_communicator_two_volumes_with_different_splits():
    mp.start_processes(_two_volumes_worker, args=(2, _free_port()), nprocs=2, join=True, start_method="spawn")


def test_launch_plan_follows_the_library_maximum():
    from tomobar_amd.slab import pd_launch_plan
    assert pd_launch_plan(30, False, kmax=3) == [3] * 10
    assert pd_launch_plan(7, False, kmax=3) == [3, 2, 2]
    assert pd_launch_plan(4, False, kmax=3) == [2, 2]
    assert pd_launch_plan(5, True, kmax=2) == [2, 2, 1]
    assert pd_launch_plan(3, False, kmax=1) == [1, 1, 1]


def test_slab_states_take_their_arrays_from_an_allocator_hook():
    """PdSlab / RofSlab ask `alloc(specs, device)` for their work arrays when one is given (the HIP drivers pass
    ops.placed_empty: views of ONE block the library places in HBM) and fall back to one allocation per array; initial
    duals are zeroed either way, and results of a run on borrowed arrays leave as copies."""
    from oracle import tomo_oracle as O
    from tomobar_amd import slab
    from tomobar_amd.slab import PdSlab, RofSlab, SlabComm, pd_tv_slab
    calls = []

    def alloc(specs, device):
        sizes = [int(np.prod(sh)) * torch.empty((), dtype=dt).element_size() for sh, dt in specs]
        block = torch.full((sum(sizes) + 64 * len(sizes),), 0xA5, dtype=torch.uint8)   # dirty memory, like a reused arena
        calls.append((len(specs), block))
        out, o = [], 0
        for (sh, dt), nb in zip(specs, sizes):
            out.append(block[o:o + nb].view(dt).view(tuple(sh)))
            o += nb + 64
        return out, None   # (arrays, lease): a private block needs no lease

    rng = np.random.default_rng(3)
    vol = (rng.random((9, 12, 20)) * 0.3).astype(np.float32)
    data = torch.from_numpy(vol)
    for half in (False, True):
        st = PdSlab(data, True, True, half, O.pd_pair_slab, O.pd_step_slab, alloc=alloc)
        n, block = calls[-1]
        assert n == 9 and st.placed
        lo, hi = block.data_ptr(), block.data_ptr() + block.numel()
        arrs = [st.inp] + st.U + st.P[0] + st.P[1]
        assert all(lo <= t.data_ptr() < hi for t in arrs)
        assert all(float(t.abs().max()) == 0.0 for t in st.P[0])                       # initial duals
        assert st.P[0][0].dtype == (torch.float16 if half else torch.float32)
        assert np.array_equal(st.local(st.inp).numpy(), vol)
    assert not PdSlab(data, False, False, False, O.pd_pair_slab, O.pd_step_slab).placed
    rs = RofSlab(data, True, False, False, O.rof_step_slab, alloc=alloc)
    assert calls[-1][0] == 3 and np.array_equal(rs.local(rs.U[0]).numpy(), vol)
    # a whole-volume run (world 1) through the driver on borrowed arrays: bit-identical, and the result is not a view of them
    keep = slab._hip_alloc
    slab._hip_alloc = alloc
    try:
        st = PdSlab(data, False, False, False, O.pd_pair_slab, O.pd_step_slab, alloc=slab._hip_alloc)
        assert st.placed
    finally:
        slab._hip_alloc = keep
    comm = SlabComm(0, 1)
    got = pd_tv_slab(data, comm, 0.04, 5, 0, 1, 8.0, False, pair_fn=O.pd_pair_slab, step_fn=O.pd_step_slab)
    assert np.array_equal(got.numpy(), O.pd_tv(vol, 0.04, 5, 0, 1, 8.0, False))
