#!/usr/bin/env python3
"""bench.py -- FISTA-OS (+PD_TV) outer iterations per second on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--strong]

N > 1: when the process was not started by a launcher (no RANK in the environment) it re-executes itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`,
one rank per GPU over RCCL (backend "nccl"); started under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  If the
node shows fewer GPUs than ranks the ranks share devices and the halo exchange is staged through the host with gloo
(a functional dry run of the multi-rank path, flagged `"oversubscribed": true`; never a performance number).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on that fits one GPU): a 3D phantom of
1024 slices of 1024^2, 900 angles over [0, pi), detector 1024 wide, ordered-subsets FISTA with 12 subsets, PD_TV
proximal step (30 inner iterations, float32 duals), non-negativity.  One "step" = one OUTER iteration = 12
sub-iterations (fused-residual forward projection, back projection with the gradient-step epilogue, 30 PD_TV inner
iterations, momentum).  Data are synthetic (ellipsoid phantom forward-projected on the GPU + Gaussian noise, seed 0)
and resident in HBM before the timed region.

Multi-GPU: the volume / sinogram are sharded into z-slabs; the projector pair is block-diagonal over z so the only
exchange is the two-plane TV halo (tomobar_amd.slab, RCCL send/recv between z-neighbours) and the scalar reductions.
  weak   (default): one 1024-slice slab PER RANK; `value` = slab-iterations/s (at N=1 exactly FISTA-OS outer
                    iterations/s of the 1024^3 problem).
  strong (--strong): the 1024 slices are split over the N ranks (balanced); `value` = outer iterations/s of the one
                    1024^3 problem.

One JSON line is printed by rank 0 (see the contract in the task statement) with two extra objects: `roofline`
(dominant kernel by time in the timed region, measured with HIP events on the launch stream by the library itself)
and `cpu_baseline` (the CPU oracle, oracle/tomo_oracle.c, on the host cores for a bounded z-subsample).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy ceiling)
FP32_VALU_PEAK_TFLOPS = 157.3  # same guide: vector f32 peak (the issue roof of the 2-tap gathers in BP / FP)

# source files whose content decides the HBM traffic of each kernel class (profiles/pmc_traffic.json is only valid
# for the sources it was measured on)
KERNEL_SOURCES = {
    "pdtv": ["tomobar_amd/csrc/tv_kernels.hip", "tomobar_amd/csrc/pd_zmarch_xk.inl", "tomobar_amd/csrc/pd_zmarch_x2.inl", "tomobar_amd/csrc/pd_zmarch2.inl"],
    "roftv": ["tomobar_amd/csrc/tv_kernels.hip", "tomobar_amd/csrc/rof_zmarch.inl"],
    "bp": ["tomobar_amd/csrc/proj_kernels.hip", "tomobar_amd/csrc/bp_brick.inl"],
    "fp": ["tomobar_amd/csrc/proj_kernels.hip", "tomobar_amd/csrc/fp_tiled.inl"],
}

# Kak-Slaney / Toft 3D head phantom on the unit cube: (A, a, b, c, x0, y0, z0, phi_deg) -- input data of the benchmark
PHANTOM_ELLIPSOIDS = [
    (1.00, 0.6900, 0.920, 0.810, 0.00, 0.0000, 0.00, 0.0),
    (-0.80, 0.6624, 0.874, 0.780, 0.00, -0.0184, 0.00, 0.0),
    (-0.20, 0.1100, 0.310, 0.220, 0.22, 0.0000, 0.00, -18.0),
    (-0.20, 0.1600, 0.410, 0.280, -0.22, 0.0000, 0.00, 18.0),
    (0.10, 0.2100, 0.250, 0.410, 0.00, 0.3500, -0.15, 0.0),
    (0.10, 0.0460, 0.046, 0.050, 0.00, 0.1000, 0.25, 0.0),
    (0.10, 0.0460, 0.046, 0.050, 0.00, -0.1000, 0.25, 0.0),
    (0.10, 0.0460, 0.023, 0.050, -0.08, -0.6050, 0.00, 0.0),
    (0.10, 0.0230, 0.023, 0.020, 0.00, -0.6060, 0.00, 0.0),
    (0.10, 0.0230, 0.046, 0.020, 0.06, -0.6050, 0.00, 0.0),
]


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--strong", action="store_true", help="strong scaling: --nz slices in total, split over the ranks")
    p.add_argument("--n", type=int, default=1024, help="slice size N = detector width")
    p.add_argument("--nz", type=int, default=1024, help="slices per GPU slab (weak) / in total (--strong)")
    p.add_argument("--angles", type=int, default=900)
    p.add_argument("--os", type=int, default=12)
    p.add_argument("--inner", type=int, default=30, help="TV inner iterations")
    p.add_argument("--reg", default="PD_TV", choices=["PD_TV", "ROF_TV", "none"])
    p.add_argument("--half", action="store_true", help="binary16 dual fields")
    p.add_argument("--ring", type=float, default=0.0, help="Group-Huber ring term: ringGH_lambda (BASELINE configs[4])")
    p.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"])
    p.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    p.add_argument("--cpu-slices", type=int, default=8)
    if len(sys.argv) == 1 and "TOMO_BENCH_ARGV" in os.environ and "RANK" in os.environ:  # rank started by self_launch()
        return p.parse_args(json.loads(os.environ["TOMO_BENCH_ARGV"]))
    return p.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
    env = dict(os.environ)
    # the ranks read their arguments from the environment: torch.distributed.run's own parser rejects options of this
    # script that are prefixes of its own (e.g. --n)
    env["TOMO_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required for RCCL between processes on this host
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def phantom_slab(n, nz_total, z_begin, nz, device):
    """Voxelised 10-ellipsoid head phantom (float32) for global slices [z_begin, z_begin+nz)."""
    import numpy as np
    import torch
    xs = ((torch.arange(n, device=device, dtype=torch.float32) - n / 2 + 0.5) / (n / 2))
    zs = ((torch.arange(z_begin, z_begin + nz, device=device, dtype=torch.float32) - nz_total / 2 + 0.5) / (nz_total / 2))
    vol = torch.zeros((nz, n, n), dtype=torch.float32, device=device)
    X = xs.view(1, 1, n)
    Y = xs.view(1, n, 1)
    Z = zs.view(nz, 1, 1)
    for A, a, b, c, x0, y0, z0, phi in PHANTOM_ELLIPSOIDS:
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


def source_hash(kernel):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES[kernel]:
        try:
            h.update(open(os.path.join(ROOT, rel), "rb").read())
        except OSError:
            h.update(b"<missing>")
    return h.hexdigest()[:16]


def cpu_baseline(args):
    """The CPU oracle (port of the reference algorithm; ASTRA / CuPy are not installable here) on the host cores:
    one FISTA-OS outer iteration on a z-subsample of the same geometry; scaled linearly in Nz (A and A^T are
    block-diagonal over z, the TV cost is linear in the voxel count)."""
    import numpy as np
    from oracle import tomo_oracle as O
    cores = O.threads()   # OpenMP team of the oracle: the CPUs this process may use (affinity mask / cgroup quota)
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
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev
    oversubscribed = world > ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    backend = None
    backend_note = None
    halo_group = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend if args.backend != "auto" else ("gloo" if oversubscribed else "nccl")
        # the default group is gloo (host scalars: barrier, timing, consensus); the halo exchange and the solver's
        # reductions run on an RCCL group when one comes up on EVERY rank, else on gloo with host-staged planes
        dist.init_process_group("gloo")
        if backend == "nccl":
            ok, why = 1, ""
            try:
                halo_group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(8, device=device)
                dist.all_reduce(probe, group=halo_group)
                ops_ = []
                peer_buf = torch.zeros(8, device=device)
                if rank + 1 < world:
                    ops_.append(dist.P2POp(dist.isend, probe, rank + 1, halo_group))
                if rank > 0:
                    ops_.append(dist.P2POp(dist.irecv, peer_buf, rank - 1, halo_group))
                for r_ in (dist.batch_isend_irecv(ops_) if ops_ else []):
                    r_.wait()
                torch.cuda.synchronize()
                if float(probe[0].item()) != float(world) or (rank > 0 and float(peer_buf[0].item()) != float(world)):
                    raise RuntimeError("RCCL self-test returned wrong data")
            except Exception as e:  # noqa: BLE001 -- any failure of the transport: fall back on every rank
                ok, why = 0, repr(e)[:200]
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                backend, halo_group = "gloo", None
                backend_note = "RCCL group failed its self-test on some rank, fell back to host-staged gloo" + (f": {why}" if why else "")
                if rank == 0:
                    print("[bench] " + backend_note, file=sys.stderr)

    from tomobar_amd import _lib
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    from tomobar_amd.slab import SlabComm, check_slab_split, slab_bounds
    lib = _lib.lib()

    n, na = args.n, args.angles
    if args.strong:
        nz_total = args.nz
        check_slab_split(nz_total, world)
        z0, z1 = slab_bounds(nz_total, world, rank)
    else:
        nz_total = args.nz * world
        z0, z1 = rank * args.nz, (rank + 1) * args.nz
    nz = z1 - z0
    angles = np.linspace(0, np.pi, na, endpoint=False)
    slab = SlabComm(rank, world, device, group=halo_group) if world > 1 else None
    rt = RecToolsIRCuPy(DetectorsDimH=n, DetectorsDimH_pad=0, DetectorsDimV=nz, CenterRotOffset=0.0, AnglesVec=angles,
                        ObjSize=n, device_projector=dev_index, OS_number=args.os)
    if slab is not None:
        rt.slab = slab
    # ---- synthetic data, resident in HBM: A(phantom) + noise
    vol = phantom_slab(n, nz_total, z0, nz, device)
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
        if args.ring > 0.0:
            d["ringGH_lambda"] = args.ring
            d["ringGH_accelerate"] = 50.0
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
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = {k: prof_read(lib, k) for k in ("pdtv", "roftv", "bp", "fp")}
    lib.tomo_profile_enable(0)
    finite = bool(torch.isfinite(out).all().item())

    if rank == 0:
        V = nz * n * n
        sub = -(-na // args.os)
        S_s = nz * sub * n
        # algorithmic bytes per unit of work; one PD_TV launch may carry two inner iterations,
        # so its bytes per launch = bytes per iteration x iterations / launches
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
                if k in ("bp", "fp"):
                    # the projectors are instruction-issue bound (2 FMAs + 2 LDS taps per voxel-angle update), far above
                    # their HBM time: report them against the f32 vector roof as well (4 flop per update)
                    tf = 4.0 * V * sub / avg / 1e9
                    kernels[k]["valu_TFLOPs"] = tf
                    kernels[k]["frac_valu"] = tf / FP32_VALU_PEAK_TFLOPS
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
        # HBM traffic per launch of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/pmc_run.sh;
        # counters cannot be collected from inside this process).  It is reported only when the kernel sources are
        # byte-identical to the ones the counters were taken on and the configuration matches.
        traffic, traffic_src = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = "pdtv_half" if (dom == "pdtv" and args.half) else dom
            if key in pmc and (n, nz) == (1024, 1024):
                ent = pmc[key]
                cur = source_hash(dom)
                traffic_src = {"measured_on_sources": ent.get("sources_sha16"), "current_sources": cur,
                               "profile": ent.get("profile")}
                if ent.get("sources_sha16") == cur:
                    traffic = ent["traffic_bytes"]
        except (OSError, ValueError):
            pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["frac_hbm"], "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": kernels[dom]["avg_ms"], "launches": kernels[dom]["launches"],
                "alg_bytes_per_launch": alg_bytes[dom]}
        if dom == "pdtv":
            # a fused PD_TV launch performs several iterations per pass through HBM: `frac` (algorithmic bytes of all its
            # iterations / time) can exceed 1; the real-traffic rate is the honest companion figure
            roof["iterations_per_launch"] = args.inner * sub_its / kernels[dom]["launches"]
        if traffic is not None:
            roof["traffic_GBps"] = traffic / kernels[dom]["avg_ms"] / 1e6
            roof["frac_traffic"] = roof["traffic_GBps"] / HBM_PEAK_GBS
        units = args.steps if args.strong else args.steps * world
        what = (f"{nz_total} slices of {n}^2 split over {world} z-slab(s)" if args.strong
                else f"{nz} slices of {n}^2 per GPU; slab-iterations/s over {world} z-slab(s)")
        ring = f"+GH ring term (lambda {args.ring:g})" if args.ring > 0.0 else ""
        line = {
            "metric": "fista_os_iterations_per_sec", "value": units / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FISTA-OS({args.os} subsets)+{args.reg}({args.inner} inner, "
                                   f"{'f16' if args.half else 'f32'} duals){ring}, {na} angles, {what} "
                                   f"(BASELINE configs[2])",
                       "slices_per_gpu": nz, "n": n, "angles": na, "os_number": args.os, "inner_iterations": args.inner,
                       "slices_per_sec": args.steps * nz_total / dt, "lipschitz_const": lc, "output_finite": finite,
                       "backend": backend, "oversubscribed": oversubscribed,
                       **({"backend_note": backend_note} if backend_note else {})},
            "roofline": roof, "kernels": kernels,
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
