#!/usr/bin/env python3
"""bench.py -- FISTA-OS (+PD_TV) outer iterations per second on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W          (N>1 is launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on that fits one GPU): a 3D phantom of
1024 slices of 1024^2, 900 angles over [0, pi), detector 1024 wide, ordered-subsets FISTA with 12 subsets, PD_TV
proximal step (30 inner iterations, float32 duals), non-negativity.  One "step" = one OUTER iteration = 12
sub-iterations (fused-residual forward projection, back projection with the gradient-step epilogue, 30 PD_TV kernel
launches, momentum).  Data are synthetic (ellipsoid phantom forward-projected on the GPU + Gaussian noise, seed 0)
and resident in HBM before the timed region.

Multi-GPU: the volume / sinogram are sharded into z-slabs, one 1024-slice slab PER RANK (weak scaling: per-GPU work
is fixed); the projector pair is block-diagonal over z so the only exchange is the one-plane TV halo
(tomobar_amd.slab, RCCL send/recv) and the scalar reductions.  `value` counts slab-iterations per second: at N=1 it is
exactly FISTA-OS outer iterations/s of the 1024^3 problem.

One JSON line is printed by rank 0 (see the contract in the task statement) with two extra objects: `roofline`
(dominant kernel by time in the timed region, measured with HIP events on the launch stream by the library itself)
and `cpu_baseline` (the CPU oracle, oracle/tomo_oracle.c, on the host cores for a bounded z-subsample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--n", type=int, default=1024, help="slice size N = detector width")
    p.add_argument("--nz", type=int, default=1024, help="slices per GPU slab")
    p.add_argument("--angles", type=int, default=900)
    p.add_argument("--os", type=int, default=12)
    p.add_argument("--inner", type=int, default=30, help="PD_TV inner iterations")
    p.add_argument("--reg", default="PD_TV", choices=["PD_TV", "ROF_TV", "none"])
    p.add_argument("--half", action="store_true", help="binary16 dual fields")
    p.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    p.add_argument("--cpu-slices", type=int, default=8)
    return p.parse_args()


def phantom_slab(n, nz_total, z_begin, nz, device):
    """Voxelised 10-ellipsoid head phantom (float32) for global slices [z_begin, z_begin+nz)."""
    from oracle.tomo_oracle import _SHEPP  # parameters only (data table)
    xs = ((torch.arange(n, device=device, dtype=torch.float32) - n / 2 + 0.5) / (n / 2))
    zs = ((torch.arange(z_begin, z_begin + nz, device=device, dtype=torch.float32) - nz_total / 2 + 0.5) / (nz_total / 2))
    vol = torch.zeros((nz, n, n), dtype=torch.float32, device=device)
    X = xs.view(1, 1, n)
    Y = xs.view(1, n, 1)
    Z = zs.view(nz, 1, 1)
    for A, a, b, c, x0, y0, z0, phi in _SHEPP:
        p = np.deg2rad(phi)
        xr = (X - x0) * float(np.cos(p)) + (Y - y0) * float(np.sin(p))
        yr = -(X - x0) * float(np.sin(p)) + (Y - y0) * float(np.cos(p))
        vol += float(A) * (((xr / a) ** 2 + (yr / b) ** 2 + ((Z - z0) / c) ** 2) <= 1.0)
    return vol


def prof_read(lib, name):
    n, ms = C.c_longlong(0), C.c_double(0.0)
    rc = lib.tomo_profile_read(name.encode(), C.byref(n), C.byref(ms))
    assert rc == 0
    return n.value, ms.value


def cpu_baseline(args):
    """The CPU oracle (port of the reference algorithm; ASTRA / CuPy are not installable here) on the host cores:
    one FISTA-OS outer iteration on a z-subsample of the same geometry; scaled linearly in Nz (A and A^T are
    block-diagonal over z, the TV cost is linear in the voxel count)."""
    from oracle import tomo_oracle as O
    cores = os.cpu_count() or 1
    nzs = args.cpu_slices
    angles = np.linspace(0, np.pi, args.angles, endpoint=False)
    P = O.Projector(nzs, args.n, args.n, angles, 0.0, args.os)
    rng = np.random.default_rng(0)
    sino = rng.random((nzs, args.angles, args.n), dtype=np.float32)
    reg = None
    if args.reg != "none":
        reg = {"method": args.reg, "regul_param": 5e-4, "iterations": args.inner, "time_marching_step": 1e-3,
               "PD_LipschitzConstant": 12.0, "methodTV": 0, "half_precision": args.half}
    t0 = time.perf_counter()
    O.fista(P, sino, 1, 2.0e4, True, reg)
    dt = time.perf_counter() - t0
    its_full = (1.0 / dt) * (nzs / args.nz)
    return {"value": its_full, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"oracle/tomo_oracle.c (OpenMP, {cores} threads): 1 outer FISTA-OS iteration on {nzs} of "
                      f"{args.nz} slices ({dt:.1f} s), scaled by {nzs}/{args.nz}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)

    from tomobar_amd import _lib
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    lib = _lib.lib()

    n, nz, na = args.n, args.nz, args.angles
    nz_total = nz * world
    angles = np.linspace(0, np.pi, na, endpoint=False)
    slab = None
    if world > 1:
        from tomobar_amd.slab import SlabComm
        slab = SlabComm(rank, world, device)
    rt = RecToolsIRCuPy(DetectorsDimH=n, DetectorsDimH_pad=0, DetectorsDimV=nz, CenterRotOffset=0.0, AnglesVec=angles,
                        ObjSize=n, device_projector=local_rank, OS_number=args.os)
    if slab is not None:
        rt.slab = slab
    # ---- synthetic data, resident in HBM: A(phantom) + noise
    vol = phantom_slab(n, nz_total, rank * nz, nz, device)
    sino = rt.Atools.forward(vol)
    gen = torch.Generator(device=device)
    gen.manual_seed(rank)
    sino += 0.01 * float(n) * torch.randn(sino.shape, generator=gen, device=device, dtype=torch.float32)
    del vol
    rt.power_seed = 0
    lc = rt.powermethod({"projection_data": None})
    reg = None
    if args.reg != "none":
        reg = {"method": args.reg, "regul_param": 5e-4, "iterations": args.inner,
               "time_marching_step": 1e-3, "half_precision": args.half}

    def run(iters):
        d = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]}
        a = {"iterations": iters, "lipschitz_const": lc, "nonnegativity": True}
        return rt.FISTA(d, a, None if reg is None else dict(reg))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    barrier()
    lib.tomo_profile_enable(1)
    t0 = time.perf_counter()
    out = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = {k: prof_read(lib, k) for k in ("pdtv", "roftv", "bp", "fp")}
    lib.tomo_profile_enable(0)
    finite = bool(torch.isfinite(out).all().item())

    if rank == 0:
        V = nz * n * n
        sub = -(-na // args.os)
        S_s = nz * sub * n
        # algorithmic bytes per unit of work; one PD_TV launch may carry two inner iterations
        # (pd_zmarch_x2), so its bytes per launch = bytes per iteration x iterations / launches
        sub_its = args.steps * args.os
        alg_total = {"pdtv": (24 if args.half else 36) * V * args.inner * sub_its, "roftv": 12 * V * args.inner * sub_its,
                     "bp": 4 * (S_s + V) * sub_its, "fp": 4 * (S_s + V) * sub_its}
        kernels = {}
        alg_bytes = {}
        for k, (cnt, ms) in prof.items():
            if cnt:
                avg = ms / cnt
                alg_bytes[k] = alg_total[k] / cnt
                kernels[k] = {"launches": cnt, "avg_ms": avg, "total_ms": ms,
                              "alg_GBps": alg_bytes[k] / avg / 1e6, "frac_hbm": alg_bytes[k] / avg / 1e6 / HBM_PEAK_GBS}
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
        # HBM traffic per launch of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/pmc_run.sh;
        # counters cannot be collected from inside this process); valid only for the configuration they were taken on
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = "pdtv_half" if (dom == "pdtv" and args.half) else dom
            if key in pmc and (n, nz) == (1024, 1024):
                traffic = pmc[key]["traffic_bytes"]
        except (OSError, ValueError):
            pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["frac_hbm"], "traffic": traffic,
                "avg_launch_ms": kernels[dom]["avg_ms"], "launches": kernels[dom]["launches"],
                "alg_bytes_per_launch": alg_bytes[dom]}
        line = {
            "metric": "fista_os_iterations_per_sec", "value": args.steps * world / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FISTA-OS({args.os} subsets)+{args.reg}({args.inner} inner, "
                                   f"{'f16' if args.half else 'f32'} duals), {na} angles, {nz} slices of {n}^2 per GPU "
                                   f"(BASELINE configs[2]); slab-iterations/s over {world} z-slab(s)",
                       "slices_per_gpu": nz, "n": n, "angles": na, "os_number": args.os, "inner_iterations": args.inner,
                       "slices_per_sec": args.steps * nz_total / dt, "lipschitz_const": lc, "output_finite": finite},
            "roofline": roof, "kernels": kernels,
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
