"""GPU test of the z-slab TV kernels (tomo_pdtv_iter_slab / tomo_roftv_iter_slab): one volume is cut into slabs that all
live on the single test GPU; the slab drivers of tomobar_amd.slab run on each of them with the ghost planes refreshed by
direct copies (what RCCL send/recv does between GPUs), and the stitched result must equal the whole-volume operator bit
for bit.  The multi-process exchange itself is covered on CPU by tests/test_slab_gloo.py."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _copy_halos(states, b):
    """rank r's send_up -> rank r+1's recv_down; rank r+1's send_down -> rank r's recv_up."""
    for r in range(len(states) - 1):
        lo, hi = states[r], states[r + 1]
        for src, dst in zip(lo.send_up(b), hi.recv_down(b)):
            dst.copy_(src)
        for src, dst in zip(hi.send_down(b), lo.recv_up(b)):
            dst.copy_(src)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("variant", [0, 22, "ranges", pytest.param(2, marks=pytest.mark.dev_variants),
                                     pytest.param(1, marks=pytest.mark.dev_variants)])
def test_pdtv_slabs_equal_whole_volume(world, half, variant):
    run_pdtv_slabs(world, half, variant)


def run_pdtv_slabs(world, half, variant, shape=(23, 37, 150), iters_list=(7, 4), methodTV=0, nonneg=1, seed=9):
    """(also driven over random shapes / splits / iteration counts by tools/fuzz_campaign.py --slabs)"""
    ranges = variant == "ranges"  # the overlapped schedule of pd_tv_slab: edge planes first, then the interior
    variant = 0 if ranges else variant
    from tomobar_amd import ops
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    from tomobar_amd.slab import PdSlab, _hip_pd_pair, _hip_pd_step, slab_bounds
    ops.set_variant("pdtv", variant)
    try:
        nz, dy, dx = shape
        rng = np.random.default_rng(seed)
        vol = (rng.random((nz, dy, dx)) * 0.3 + (np.indices((nz, dy, dx))[2] > dx // 2) - 0.5).astype(np.float32)
        vd = torch.from_numpy(vol).cuda()
        tau = np.float32(0.04 * 0.1)
        sigma = np.float32(1.0 / (8.0 * tau))
        lt = np.float32(tau / 0.04)
        for iters in iters_list:  # pairs + an odd trailing iteration / pairs only
            want = PD_TV_cupy(vd, 0.04, iters, methodTV, nonneg, 8.0, 0, half).cpu().numpy()
            states = []
            for r in range(world):
                z0, z1 = slab_bounds(nz, world, r)
                states.append(PdSlab(vd[z0:z1].contiguous(), r > 0, r < world - 1, half, _hip_pd_pair, _hip_pd_step))
            for r in range(world - 1):  # the exchange before the first step: two planes of Input either side
                lo, hi = states[r], states[r + 1]
                for src, dst in zip(lo.initial_send_up(), hi.initial_recv_down()):
                    dst.copy_(src)
                for src, dst in zip(hi.initial_send_down(), lo.initial_recv_up()):
                    dst.copy_(src)
            from tomobar_amd.slab import pd_launch_plan
            args = (sigma, tau, lt, np.float32(1.0), methodTV, nonneg)
            plan = pd_launch_plan(iters, half)   # the launches tomo_pdtv itself makes: 3 + 3 + ... (2 for binary16 duals)
            for n, k in enumerate(plan):
                if ranges and k >= 2:
                    for s in states:
                        for z0, z1 in s.boundary_ranges()[0]:
                            s.multi_range(k, *args, z0, z1)
                    _copy_halos(states, states[0].cur ^ 1)  # "in flight" while the interiors are computed
                    for s in states:
                        b0, b1 = s.boundary_ranges()[1]
                        s.multi_range(k, *args, b0, b1)
                        s.flip()
                    continue
                for s in states:
                    if k >= 2:
                        s.multi(k, *args)
                    else:
                        s.single(*args)
                _copy_halos(states, states[0].cur)
            got = torch.cat([s.result() for s in states]).cpu().numpy()
            assert np.array_equal(got, want), (shape, world, iters, np.abs(got - want).max())
    finally:
        ops.set_variant("pdtv", 0)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("half", [False, True])
def test_roftv_slabs_equal_whole_volume(world, half):
    run_roftv_slabs(world, half)


def run_roftv_slabs(world, half, shape=(19, 21, 90), iters=6, seed=10):
    from tomobar_amd.regularisersCuPy import ROF_TV_cupy
    from tomobar_amd.slab import RofSlab, _hip_rof_step, slab_bounds
    nz, dy, dx = shape
    rng = np.random.default_rng(seed)
    vol = (rng.random((nz, dy, dx)) * 0.3 + (np.indices((nz, dy, dx))[2] > dx // 2)).astype(np.float32)
    vd = torch.from_numpy(vol).cuda()
    want = ROF_TV_cupy(vd, 0.05, iters, 0.005, 0, half).cpu().numpy()
    states = []
    for r in range(world):
        z0, z1 = slab_bounds(nz, world, r)
        states.append(RofSlab(vd[z0:z1].contiguous(), r > 0, r < world - 1, half, _hip_rof_step))
    _copy_halos(states, 0)
    for it in range(iters):
        if it % 2:  # every other iteration in the overlapped order: boundary planes, "exchange", interior
            for s in states:
                for zr in s.boundary_ranges()[0]:
                    s.step(it, np.float32(0.05), np.float32(0.005), zr)
            _copy_halos(states, (it + 1) & 1)
            for s in states:
                s.step(it, np.float32(0.05), np.float32(0.005), s.boundary_ranges()[1])
            continue
        for s in states:
            s.step(it, np.float32(0.05), np.float32(0.005))
        _copy_halos(states, (it + 1) & 1)
    got = torch.cat([s.local(s.U[iters & 1]) for s in states]).cpu().numpy()
    assert np.array_equal(got, want), (shape, world, iters, np.abs(got - want).max())


def test_halo_pack_unpack_round_trip():
    """tomo_halo_pack / tomo_halo_unpack: blocks of different element sizes, odd byte counts and unaligned starts gathered
    into one staging buffer and scattered back, byte for byte; the staging layout is the documented one (16-byte aligned
    block starts)."""
    import ctypes as C
    from tomobar_amd import _lib
    from tomobar_amd.slab import _halo_staging_bytes, _hip_halo_pack, _hip_halo_unpack
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    u = torch.rand((7, 13, 37), device="cuda", generator=g)                       # float32 planes, 13*37*4 = 1924 B each
    p = torch.rand((7, 13, 37), device="cuda", generator=g).to(torch.float16)     # binary16 planes, 962 B each (not 4-aligned)
    b = torch.randint(0, 255, (1001,), device="cuda", dtype=torch.uint8, generator=g)
    blocks = [u[4:7], p[4:7], p[1:2], b[3:1000], u[0:1]]
    nbytes = [t.numel() * t.element_size() for t in blocks]
    total = _halo_staging_bytes(nbytes)
    arr = (C.c_size_t * len(nbytes))(*nbytes)
    assert _lib.lib().tomo_halo_staging_bytes(arr, len(nbytes)) == total
    stage = torch.full((total,), 0xAB, dtype=torch.uint8, device="cuda")
    _hip_halo_pack(blocks, nbytes, stage)
    off = 0
    for t, nb in zip(blocks, nbytes):
        assert torch.equal(stage[off:off + nb], t.contiguous().view(-1).view(torch.uint8))
        off += (nb + 15) // 16 * 16
    outs = [torch.zeros_like(t.contiguous()) for t in blocks]
    _hip_halo_unpack(stage, outs, nbytes)
    torch.cuda.synchronize()
    for t, o in zip(blocks, outs):
        assert torch.equal(t, o)
    with pytest.raises(ValueError):
        _hip_halo_pack([u[0:1]] * 9, [4] * 9, stage)   # more than 8 blocks per call


def test_pdtv_thin_volumes_direct_abi():
    """tomo_pdtv on 3D volumes thinner than a fused launch's iteration count (dz = 1, 2) takes
    fewer iterations per launch; checked against the oracle's 3D kernel on the same un-squeezed shape."""
    from oracle import tomo_oracle as O
    from tomobar_amd import ops
    for variant in (0, 22):   # as shipped (relaxed float32 arithmetic: tolerance) / the reference's roundings (bit for bit)
        ops.set_variant("pdtv", variant)
        for dz in (1, 2):
            rng = np.random.default_rng(dz)
            x = (rng.random((dz, 9, 70)) * 0.3 + (np.indices((dz, 9, 70))[2] > 35)).astype(np.float32)
            sigma, tau, lt, theta = O.pd_scalars(0.04, 8.0)
            want = np.empty_like(x)
            assert O.lib().orc_pdtv(O._fptr(x), O._fptr(want), 70, 9, dz, 3, sigma, tau, lt, theta, 7, 0, 1, 0) == 0
            xd = torch.from_numpy(x).cuda()
            got = torch.empty_like(xd)
            ops.pdtv(xd, got, sigma, tau, lt, theta, 7, 0, 1, False)
            torch.cuda.synchronize()
            g = got.cpu().numpy()
            err = np.linalg.norm(g - want) / np.linalg.norm(want)
            assert err < 1e-6 and (variant != 22 or np.array_equal(g, want)), (variant, dz, err)
    ops.set_variant("pdtv", 0)


def test_placed_work_arrays_are_disjoint_views_of_one_library_block():
    """ops.placed_empty (the slab drivers' allocator): one tensor per spec inside one block the library owns and places,
    256-byte aligned, ARRAY_SKEW apart, writable without touching each other; the same request returns the same block;
    pd_tv_slab on such arrays returns a copy (the block is reused by the next call)."""
    from tomobar_amd import ops
    from tomobar_amd.slab import SlabComm, pd_tv_slab
    dev = torch.device("cuda", 0)
    specs = [((5, 33, 70), torch.float32)] * 3 + [((5, 33, 70), torch.float16)] * 2
    a, lease_a = ops.placed_empty(specs, dev, slot=3)
    assert ops.lease_is_current(lease_a)
    assert [tuple(t.shape) for t in a] == [s for s, _ in specs] and [t.dtype for t in a] == [d for _, d in specs]
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in a)
    for (b0, e0), (b1, e1) in zip(spans, spans[1:]):
        assert b0 % 256 == 0 and b1 % 256 == 0 and b1 - e0 >= ops.ARRAY_SKEW
    for i, t in enumerate(a):
        t.fill_(i + 1)
    torch.cuda.synchronize()
    assert [float(t.float().min()) for t in a] == [float(t.float().max()) for t in a] == [1.0, 2.0, 3.0, 4.0, 5.0]
    b, lease_b = ops.placed_empty(specs, dev, slot=3)
    assert [t.data_ptr() for t in b] == [t.data_ptr() for t in a]
    assert ops.lease_is_current(lease_b) and not ops.lease_is_current(lease_a), "a second request supersedes the first lease"
    # the driver's result must survive the next call on the same block
    rng = np.random.default_rng(2)
    v1 = torch.from_numpy(rng.random((9, 40, 70)).astype(np.float32)).cuda()
    v2 = torch.from_numpy(rng.random((9, 40, 70)).astype(np.float32)).cuda()
    comm = SlabComm(0, 1, dev)
    r1 = pd_tv_slab(v1, comm, 0.04, 6, 0, 1, 8.0, False)
    keep = r1.clone()
    r2 = pd_tv_slab(v2, comm, 0.04, 6, 0, 1, 8.0, False)
    assert torch.equal(r1, keep) and not torch.equal(r1, r2)
    # PD_TV and ROF_TV solvers take different blocks (slots 0 / 1): running one never supersedes the other's lease ...
    from tomobar_amd.slab import PLACED_SLOT_PD, PLACED_SLOT_ROF, PdSlab, _check_lease, _hip_alloc, rof_tv_slab
    st = PdSlab(v1, False, False, False, None, None, alloc=_hip_alloc(PLACED_SLOT_PD))
    rof_tv_slab(v2, comm, 0.04, 3, 0.002, False)
    _check_lease(st)
    assert PLACED_SLOT_PD != PLACED_SLOT_ROF
    # ... a second PD_TV solver on the same stream does, and the first one refuses to hand out a result afterwards
    pd_tv_slab(v2, comm, 0.04, 3, 0, 1, 8.0, False)
    with pytest.raises(RuntimeError, match="another solver"):
        _check_lease(st)
