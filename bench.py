#!/usr/bin/env python3
"""bench.py -- FISTA-OS (+PD_TV) outer iterations per second on MI355X, BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W [--config cfg1|cfg2|cfg3|cfg3-share|cfg5|cfg5-share] [--strong]

N > 1: when the process was not started by a launcher (no RANK in the environment) it re-executes itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`,
one rank per GPU over RCCL (backend "nccl"); started under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.  If the
node shows fewer GPUs than ranks the ranks share devices and the halo exchange is staged through the host with gloo
(a functional dry run of the multi-rank path, flagged `"oversubscribed": true`; never a performance number).

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on that fits one GPU): a 3D phantom of
1024 slices of 1024^2, 900 angles over [0, pi), detector 1024 wide, ordered-subsets FISTA with 12 subsets, PD_TV
proximal step (30 inner iterations, float32 duals), non-negativity.  One "step" = one OUTER iteration = 12
sub-iterations (fused-residual forward projection, back projection with the gradient-step epilogue, 30 PD_TV inner
iterations, momentum).  Data are synthetic and resident in HBM before the timed region: the ANALYTIC line integrals of
the ellipsoid phantom, evaluated on the GPU, with Poisson noise.  `--config` selects the other BASELINE
configurations (PRESETS below); explicit --n/--nz/... flags override a preset and the workload string always states
what actually ran.

Multi-GPU: the volume / sinogram are sharded into z-slabs; the projector pair is block-diagonal over z so the only
exchange is the three-plane TV halo (tomobar_amd.slab: packed, one RCCL send/recv per neighbour and launch) and the
scalar reductions.
  weak   (default): one --nz-slice slab PER RANK; `value` = slab-iterations/s (at N=1 exactly outer iterations/s).
  strong (--strong): the --nz slices are split over the N ranks (balanced); `value` = outer iterations/s of the ONE
                    problem -- `--strong --config cfg5` is the north-star workload (2560^2 x 2160, 1800 angles).
Per-rank exchange statistics (messages, bytes, host time posting / waiting, compute-stream stall) are in `halo`.

One JSON line is printed by rank 0 (see the contract in the task statement) with extra objects: `roofline`
(dominant kernel by time in the timed region, measured with HIP events on the launch stream by the library itself),
`roofline_bp` / `roofline_fp` (the two projectors whatever dominates) and `cpu_baseline` (the CPU oracle, oracle/tomo_oracle.c, on the host cores for a z-subsample of the SAME sinogram).
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
LDS_PEAK_BPS = 256.0 * 256 * 2.4e9  # same guide, LDS: ds_read_b128 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 TB/s
SUSTAINED_CLOCK_HZ = 2.03e9         # what the chip holds under the projector kernels (GRBM_GUI_ACTIVE / wall time,
                                    # docs/kernels/bp.md): the LDS fractions are reported against both clocks

# source files whose content decides the HBM traffic of each kernel class (profiles/pmc_traffic.json is only valid
# for the sources it was measured on)
KERNEL_SOURCES = {
    "pdtv": ["tomobar_amd/csrc/tv_kernels.hip", "tomobar_amd/csrc/pd_zmarch_xk.inl", "tomobar_amd/csrc/pd_zmarch_x2.inl", "tomobar_amd/csrc/pd_zmarch2.inl",
             "tomobar_amd/csrc/pd_rows2d.inl"],
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


# BASELINE.json configs[1..4] (configs[0] is the CPU plumbing case, tests/test_cfg1_cpu.py).  "-share" = the slab ONE GPU
# holds when the configuration is sharded over the number of GPUs BASELINE names (4 for configs[3], 8 for configs[4]).
PRESETS = {
    "cfg1": dict(n=256, nz=256, angles=360, os=1, reg="none", method="FISTA", baseline="configs[1]"),
    "cfg2": dict(n=1024, nz=1024, angles=900, os=12, reg="PD_TV", inner=30, method="FISTA", baseline="configs[2]"),
    "cfg3": dict(n=2048, nz=1024, angles=1500, os=1, reg="ROF_TV", inner=20, method="ADMM", baseline="configs[3]"),
    "cfg3-share": dict(n=2048, nz=256, angles=1500, os=1, reg="ROF_TV", inner=20, method="ADMM",
                       baseline="configs[3], one GPU's slab of 4"),
    "cfg5": dict(n=2560, nz=2160, angles=1800, os=12, reg="PD_TV", inner=30, ring=1e-4, method="FISTA",
                 baseline="configs[4]"),
    "cfg5-share": dict(n=2560, nz=270, angles=1800, os=12, reg="PD_TV", inner=30, ring=1e-4, method="FISTA",
                       baseline="configs[4], one GPU's slab of 8"),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default=None, choices=sorted(PRESETS), help="BASELINE configuration preset (default cfg2)")
    p.add_argument("--strong", action="store_true", help="strong scaling: --nz slices in total, split over the ranks")
    p.add_argument("--n", type=int, default=None, help="slice size N = detector width")
    p.add_argument("--nz", type=int, default=None, help="slices per GPU slab (weak) / in total (--strong)")
    p.add_argument("--angles", type=int, default=None)
    p.add_argument("--os", type=int, default=None)
    p.add_argument("--inner", type=int, default=None, help="TV inner iterations")
    p.add_argument("--reg", default=None, choices=["PD_TV", "ROF_TV", "none"])
    p.add_argument("--method", default=None, choices=["FISTA", "ADMM"])
    p.add_argument("--half", action="store_true", help="binary16 dual fields")
    p.add_argument("--ring", type=float, default=None, help="Group-Huber ring term: ringGH_lambda (BASELINE configs[4])")
    p.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"])
    p.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    p.add_argument("--no-pmc", action="store_true",
                   help="never spawn the live rocprofv3 --pmc passes that measure roofline.traffic")
    p.add_argument("--live-pmc", action="store_true",
                   help="N = 1: measure roofline.traffic live with two child rocprofv3 --pmc passes (~25 s; each re-allocates the "
                        "dominant kernel's arrays while this process keeps its own).  On by default for the default workload "
                        "only, where it is known to fit; BENCH_LIVE_PMC=1 acts like the flag")
    p.add_argument("--cpu-slices", type=int, default=8)
    p.add_argument("--exact-tv", action="store_true",
                   help="PD_TV float32 duals with the reference's rounding sequence (tomo_set_variant('pdtv', 22): bit-identical "
                        "to the oracle, +5 ... +16 %% per launch; the default is within 1e-5); the workload string says so")
    p.add_argument("--north-star", action="store_true",
                   help="N > 1: after the headline workload also run strong scaling of BASELINE configs[4] as a SUPERVISED CHILD JOB "
                        "(a second torch.distributed.run started by rank 0 once every rank has finished and the headline is on "
                        "stderr; hard wall-clock limit BENCH_NORTH_STAR_TIMEOUT, default 900 s) and report it as an extra "
                        "`north_star` block.  On by default from 4 ranks on real GPUs (where every rank's share fits); "
                        "BENCH_NORTH_STAR=1 acts like the flag, BENCH_NORTH_STAR=0 / --no-north-star switch it off")
    p.add_argument("--no-north-star", action="store_true", help="never run the north-star child job")
    if len(sys.argv) == 1 and "TOMO_BENCH_ARGV" in os.environ and "RANK" in os.environ:  # rank started by self_launch()
        args = p.parse_args(json.loads(os.environ["TOMO_BENCH_ARGV"]))
    else:
        args = p.parse_args()
    # A driver that only varies --gpus can still reach the other workloads: BENCH_CONFIG=<preset> and BENCH_STRONG=1 act
    # like --config / --strong when the flags are absent (README.md, "Benchmark").
    if args.config is None and os.environ.get("BENCH_CONFIG"):
        if os.environ["BENCH_CONFIG"] not in PRESETS:
            p.error(f"BENCH_CONFIG={os.environ['BENCH_CONFIG']!r}: choose from {sorted(PRESETS)}")
        args.config = os.environ["BENCH_CONFIG"]
    if os.environ.get("BENCH_STRONG", "0") not in ("", "0"):
        args.strong = True
    if os.environ.get("BENCH_NORTH_STAR", "0") not in ("", "0") or os.environ.get("BENCH_NORTH_STAR_TEST", "0") not in ("", "0"):
        args.north_star = True
    if os.environ.get("BENCH_NORTH_STAR", "") == "0":
        args.no_north_star = True
    if os.environ.get("BENCH_LIVE_PMC", "0") not in ("", "0"):
        args.live_pmc = True
    apply_preset(args)
    return args


def apply_preset(args):
    """Fill the unset workload flags from the preset (default cfg2) and remember what was overridden."""
    preset = dict(PRESETS[args.config or "cfg2"])
    args.baseline = preset.pop("baseline")
    overridden = []
    for key, val in {**dict(inner=30, ring=0.0), **preset}.items():
        if getattr(args, key) is None:
            setattr(args, key, val)
        elif getattr(args, key) != val and key in preset:
            overridden.append(key)
    if args.half:
        overridden.append("half")
    if getattr(args, "exact_tv", False):
        overridden.append("PD_TV with the reference's roundings (variant 22)")
    args.overridden = overridden   # the workload string names the BASELINE config only when nothing was overridden


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


def analytic_sinogram(n, nz_total, z_begin, nz, angles, device):
    """Line integrals of the ellipsoid phantom on the engine's geometry, evaluated on the GPU: [nz, na, n] float32 in voxel
    units (analytic ellipsoid sinogram; same formula as oracle.shepp_logan_sino).  torch is plumbing here:
    this is input data, not part of the measured path."""
    import numpy as np
    import torch
    na = len(angles)
    th = torch.as_tensor(np.asarray(angles), dtype=torch.float64, device=device).view(1, na, 1)
    s = ((torch.arange(n, device=device, dtype=torch.float64) - n / 2 + 0.5) / (n / 2)).view(1, 1, n)
    out = torch.empty((nz, na, n), dtype=torch.float32, device=device)
    for zb in range(0, nz, 32):                                                     # bounded float64 temporaries
        ze = min(zb + 32, nz)
        zs = ((torch.arange(z_begin + zb, z_begin + ze, device=device, dtype=torch.float64) - nz_total / 2 + 0.5)
              / (nz_total / 2)).view(ze - zb, 1, 1)
        acc = torch.zeros((ze - zb, na, n), dtype=torch.float64, device=device)
        for A, a, b, c, x0, y0, z0, phi in PHANTOM_ELLIPSOIDS:
            p = float(np.deg2rad(phi))
            k2 = torch.clamp(1.0 - ((zs - z0) / c) ** 2, min=0.0)                   # section scale^2, [nz,1,1]
            r2 = (a * torch.cos(th - p)) ** 2 + (b * torch.sin(th - p)) ** 2        # [1,na,1]
            d = s - (x0 * torch.cos(th) + y0 * torch.sin(th))                       # [1,na,n]
            disc = torch.clamp(r2 * k2 - d * d, min=0.0)
            acc += A * 2.0 * a * b * torch.sqrt(disc) / r2
        out[zb:ze] = (acc * (n / 2)).to(torch.float32)
    return out


def cpu_baseline(args, sino_dev, lc):
    """The CPU oracle (port of the reference algorithm; ASTRA / CuPy are not installable here) on the host cores: one
    outer iteration of the same loop on a z-subsample of the SAME sinogram with the same Lipschitz constant, scaled
    linearly in Nz (A and A^T are block-diagonal over z, the TV cost is linear in the voxel count)."""
    import numpy as np
    from oracle import tomo_oracle as O
    cores = O.threads()   # OpenMP team of the oracle: the CPUs this process may use (affinity mask / cgroup quota)
    nz = int(sino_dev.shape[0])
    nzs = min(args.cpu_slices, nz)
    zsel = np.unique(np.linspace(0, nz - 1, nzs).round().astype(int))
    nzs = len(zsel)
    sino = np.ascontiguousarray(sino_dev[zsel.tolist()].cpu().numpy())
    angles = np.linspace(0, np.pi, args.angles, endpoint=False)
    P = O.Projector(nzs, args.n, args.n, angles, 0.0, args.os)
    reg = None
    if args.reg != "none":
        reg = {"method": args.reg, "regul_param": 5e-4, "iterations": args.inner, "time_marching_step": 1e-3,
               "PD_LipschitzConstant": 12.0, "methodTV": 0, "half_precision": args.half}
    t0 = time.perf_counter()
    if args.method == "ADMM":
        O.admm(P, sino, 1, lc, 1.0, 1.6, True, reg)
    elif args.ring > 0.0:
        O.fista(P, sino, 1, lc, True, reg, ring={"lambda": args.ring, "accelerate": 50})
    else:
        O.fista(P, sino, 1, lc, True, reg)
    dt = time.perf_counter() - t0
    its_full = (1.0 / dt) * (nzs / nz)
    return {"value": its_full, "unit": "iterations/s", "cores": cores, "kind": "port",
            "sample": f"oracle/tomo_oracle.c (OpenMP, {cores} threads): 1 outer {args.method} iteration of an {nzs}-slice 3D problem "
                      f"({nzs} of the {nz} slices of the GPU leg's own sinogram, spread over z; the 3D TV prox sees {nzs} planes, i.e. "
                      f"two z-faces per {nzs} planes instead of per {nz}) in {dt:.1f} s, scaled by {nzs}/{nz}"}


def live_traffic(dom, args, n, nz, sub):
    """HBM traffic per launch of the dominant kernel, measured NOW on this box: two separate `rocprofv3 --pmc` passes
    (FETCH_SIZE, then WRITE_SIZE; kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes) over
    tools/pmc_probe.py, which launches that kernel on the bench's own shape in a child process.  FETCH_SIZE is doubled (the
    guide's gfx950 correction: 128-B requests are tallied as 64 B; calibrated on a dword stream in profiles/archive/r1_pdtv_pmc.txt),
    values are KiB.  Returns (bytes per launch, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process already runs under a profiler: no nested counter passes"
    quad = args.ring <= 0.0   # the drivers hand the residual over quad-interleaved unless a ring term reads it
    what = {"pdtv": ("pdtv22" if getattr(args, "exact_tv", False) else "pdtv0") + ("h" if args.half else ""),
            "roftv": "roftv", "bp": "bpq" if quad else "bp0", "fp": "fpq" if quad else "fp"}[dom]
    key = {"pdtv": ["xk_kernel"], "roftv": ["rof_"], "bp": ["bp_brick"], "fp": ["fp_tiled", "transpose"]}[dom]
    try:
        tmp = tempfile.mkdtemp(prefix="tomo_pmc_", dir="/tmp")
    except OSError as e:
        return None, f"no scratch directory for the counter passes: {e!r}"[:160]
    env = dict(os.environ, TMPDIR="/tmp", PMC_PD_ITERS="9")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py"), what, str(n), str(nz), str(sub)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            rows = [row for row in csv.DictReader(open(files[0])) if row.get("Counter_Name") == counter]
            got[counter] = [[float(row["Counter_Value"]) for row in rows if k in row["Kernel_Name"]] for k in key]
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return None, f"counter pass failed: {e!r}"[:160]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kib = lambda f, w: (2.0 * f + w) * 1024.0  # noqa: E731
    try:
        if dom == "pdtv":      # a 9-iteration prox = first (no dual reads) + middle + last (no dual stores) launch
            f, w = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
            tr = [kib(a, b) for a, b in zip(f, w)]
            assert len(tr) == 3
            plan_launches = max(1, -(-args.inner // 3))
            mean = (tr[0] + tr[2] + (plan_launches - 2) * tr[1]) / plan_launches if plan_launches >= 2 else tr[1]
            return mean, (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/pmc_probe.py {what} {n} {nz} on this box; "
                          f"first / middle / last launch {tr[0]:.4g} / {tr[1]:.4g} / {tr[2]:.4g} B, mean of the {plan_launches} launches of a prox")
        if dom == "fp":        # one call = two stepping-axis launches + the in-plane transpose; the probe makes two calls
            fs = sum(sum(v) for v in got["FETCH_SIZE"]) / 2.0
            ws = sum(sum(v) for v in got["WRITE_SIZE"]) / 2.0
            return kib(fs, ws), f"live: rocprofv3 --pmc passes over tools/pmc_probe.py fp {n} {nz} {sub}; per forward projection (its stepping-class launches + the in-plane transpose)"
        f, w = got["FETCH_SIZE"][0][-1], got["WRITE_SIZE"][0][-1]   # steady-state launch: the last one
        return kib(f, w), f"live: rocprofv3 --pmc passes over tools/pmc_probe.py {what} {n} {nz} {sub}; last launch"
    except (AssertionError, IndexError) as e:
        return None, f"unexpected counter rows: {e!r}"


def footprint_bytes(args, nz):
    """HBM one rank needs for a slab of nz slices (estimate, used only to decide whether the north-star block fits):
    solver volumes (X, X_t, X_old, transposed copy, prox output) + TV scratch (2 U + 6 duals, or 2 U for ROF) + the
    sinogram and one subset's residual, + the torch temporaries of the data synthesis (3 sinograms)."""
    v = 4.0 * nz * args.n * args.n
    s_full = 4.0 * nz * args.angles * args.n
    tv = {"PD_TV": 2 * v + 6 * (v / 2 if args.half else v), "ROF_TV": 2 * v, "none": 0.0}[args.reg]
    vols = (6 if args.method == "ADMM" else 5) * v
    return vols + tv + s_full * (1.0 + 1.0 / max(args.os, 1)) + 3.0 * s_full


def measure(args, env):
    """One workload: data synthesis, warm-up, the timed region, the JSON line (rank 0; None elsewhere)."""
    import numpy as np
    import torch
    rank, world, device, dev_index = env["rank"], env["world"], env["device"], env["dev_index"]
    dist, halo_group, backend, backend_note = env["dist"], env["halo_group"], env["backend"], env["backend_note"]
    oversubscribed = env["oversubscribed"]
    from tomobar_amd import _lib
    from tomobar_amd.methodsIR_CuPy import RecToolsIRCuPy
    from tomobar_amd.slab import GHOST, SlabComm, check_slab_split, pd_launch_plan, slab_bounds
    lib = _lib.lib()
    from tomobar_amd import ops as _ops
    _ops.set_variant("pdtv", 22 if getattr(args, "exact_tv", False) else 0)

    n, na = args.n, args.angles
    if args.strong:
        nz_total = args.nz
        # every rank evaluates the same test before anything is posted: 3D PD_TV keeps GHOST-plane ghosts, ROF_TV two
        check_slab_split(nz_total, world, min_slices=GHOST if args.reg == "PD_TV" else 2)
        z0, z1 = slab_bounds(nz_total, world, rank)
    else:
        nz_total = args.nz * world
        z0, z1 = rank * args.nz, (rank + 1) * args.nz
    nz = z1 - z0
    angles = np.linspace(0, np.pi, na, endpoint=False)
    slab = SlabComm(rank, world, device, group=halo_group) if world > 1 else None
    rt = RecToolsIRCuPy(DetectorsDimH=n, DetectorsDimH_pad=0, DetectorsDimV=nz, CenterRotOffset=0.0, AnglesVec=angles,
                        ObjSize=n, device_projector=dev_index, OS_number=args.os if args.os > 1 else None)
    if slab is not None:
        slab.timing = True
        rt.slab = slab
    if args.reg != "none" and slab is None and os.environ.get("BENCH_RESERVE_LATE", "0") in ("", "0"):
        # set-up, like the context and the sinogram: the library allocates and PLACES its TV scratch arena now (a search
        # over up to eight candidate blocks, 0.1-6 s once per process; docs/kernels/placement.md) -- FIRST, while the device
        # is still empty: the search can hold more candidates and sees unfragmented memory.  RecToolsIRCuPy does the same at
        # the start of its first call (then a no-op here); done explicitly so that --warmup 0 does not time it
        from tomobar_amd import ops as _ops0
        _ops0.reserve_tv_scratch((nz, n, n), device, args.reg, args.half)
    # ---- synthetic data, resident in HBM: analytic line integrals of the ellipsoid phantom, transmission
    #      Poisson noise (I0 = 2e4 photons per pixel, peak attenuation 3), seed = rank
    sino = analytic_sinogram(n, nz_total, z0, nz, angles, device)
    peak = float(sino.max().item())
    if dist is not None:
        t = torch.tensor([peak], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        peak = float(t.item())
    mu, i0 = 3.0 / max(peak, 1e-6), 2.0e4
    gen = torch.Generator(device=device)
    gen.manual_seed(rank)
    counts = torch.poisson(i0 * torch.exp(-mu * sino), generator=gen).clamp_(min=1.0)
    sino = (-(torch.log(counts / i0)) / mu).to(torch.float32).contiguous()
    del counts
    rt.power_seed = 0
    lc = rt.powermethod({"projection_data": None})
    reg = None
    if args.reg != "none":
        reg = {"method": args.reg, "regul_param": 5e-4, "iterations": args.inner,
               "time_marching_step": 1e-3, "half_precision": args.half}

    if reg is not None and slab is None:
        _ops.reserve_tv_scratch((nz, n, n), device, args.reg, args.half)   # (a no-op unless BENCH_RESERVE_LATE=1 skipped the early one)

    def run(iters):
        d = {"projection_data": sino, "data_axes_labels_order": ["detY", "angles", "detX"]}
        if args.ring > 0.0:
            d["ringGH_lambda"] = args.ring
            d["ringGH_accelerate"] = 50.0
        a = {"iterations": iters, "lipschitz_const": lc, "nonnegativity": True}
        if args.method == "ADMM":
            a.update({"ADMM_rho_const": 1.0, "ADMM_relax_par": 1.6})
            return rt.ADMM(d, a, None if reg is None else dict(reg))
        return rt.FISTA(d, a, None if reg is None else dict(reg))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    barrier()
    if slab is not None:
        slab.timing_summary()                      # drop the warm-up's events
        for k in ("exchanges", "messages", "bytes"):
            slab.stats[k] = 0
        for k in ("post_host_ms", "wait_host_ms"):
            slab.stats[k] = 0.0
        slab._wait_stream_ms = 0.0
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
    halo = None
    if dist is not None:
        mine = dict(rank=rank, slices=nz, **slab.timing_summary())
        halo = [None] * world
        dist.all_gather_object(halo, mine)

    if rank == 0:
        V = nz * n * n
        sub = -(-na // args.os)
        S_s = nz * sub * n
        sub_its = args.steps * args.os
        # SURVEY 8d per-iteration figures (bytes one inner iteration / one projector call must move)
        pd_it = (24 if args.half else 36) * V
        alg_total = {"pdtv": pd_it * args.inner * sub_its, "roftv": 12 * V * args.inner * sub_its,
                     "bp": 4 * (S_s + V) * sub_its, "fp": 4 * (S_s + V) * sub_its}
        # COMPULSORY bytes of the launches as executed: a fused PD_TV launch reads U, P1..3, Input and writes U, P1..3
        # ONCE for its k iterations (the first launch of a prox reads no duals, the last stores none), so the bytes a
        # launch must move do not grow with k -- the per-iteration figure x k is an equivalent rate, not a roofline
        comp_total = dict(alg_total)
        if args.reg == "PD_TV":
            plan = pd_launch_plan(args.inner, args.half)
            pb = 2 if args.half else 4
            per_prox = 0
            flags = world == 1   # the whole-volume driver tells a three-iteration launch when the duals are zero / unused
            for i, k in enumerate(plan):
                first = flags and i == 0 and k == 3   # U IS Input then (one array), and the duals are zero (not read)
                rd = (4 if first else 4 + 4 + 3 * pb)                                   # U, Input, P1..3
                wr = 4 + (0 if (flags and i == len(plan) - 1 and k == 3) else 3 * pb)  # U, P1..3
                per_prox += (rd + wr) * V
            comp_total["pdtv"] = per_prox * sub_its
        kernels = {}
        for k, (cnt, ms) in prof.items():
            if cnt:
                avg = ms / cnt
                comp = comp_total[k] / cnt
                kernels[k] = {"launches": cnt, "avg_ms": avg, "total_ms": ms,
                              "compulsory_bytes_per_launch": comp,
                              "GBps": comp / avg / 1e6, "frac_hbm": comp / avg / 1e6 / HBM_PEAK_GBS}
                if k == "pdtv":
                    kernels[k]["iterations_per_launch"] = args.inner * sub_its / cnt
                    kernels[k]["alg_GBps_per_iteration_equiv"] = alg_total[k] / cnt / avg / 1e6
                if k in ("bp", "fp"):
                    # the projectors are instruction-issue bound (2 FMAs + 2 LDS taps per voxel-angle update), far above
                    # their HBM time: report them against the f32 vector roof (4 flop per update) and against the LDS
                    # floor (8 B of ds_read_b128 per update at 256 B/clk/CU x 256 CUs x 2.4 GHz) as well
                    upd = float(V) * sub
                    tf = 4.0 * upd / avg / 1e9
                    kernels[k]["valu_TFLOPs"] = tf
                    kernels[k]["frac_valu"] = tf / FP32_VALU_PEAK_TFLOPS
                    kernels[k]["updates_per_s"] = upd / avg * 1e3
                    kernels[k]["lds_floor_ms"] = upd * 8.0 / LDS_PEAK_BPS * 1e3           # at the 2.4 GHz spec clock
                    kernels[k]["frac_lds"] = kernels[k]["lds_floor_ms"] / avg
                    kernels[k]["lds_floor_ms_sustained"] = kernels[k]["lds_floor_ms"] * 2.4e9 / SUSTAINED_CLOCK_HZ
                    kernels[k]["frac_lds_sustained"] = kernels[k]["lds_floor_ms_sustained"] / avg
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"])
        # HBM traffic per launch of the dominant kernel comes from separate rocprofv3 --pmc passes (tools/pmc_run.sh;
        # counters cannot be collected from inside this process).  It is reported only when the kernel sources are
        # byte-identical to the ones the counters were taken on and the configuration matches.
        default_shape = (n, nz, na, args.os) == (1024, 1024, 900, 12)

        def committed_traffic(kern):
            """(bytes per launch, provenance) from profiles/pmc_traffic.json -- only for the shape it was measured on and only
            while the kernel sources still hash to what the counters were taken on (tests/test_host_logic.py keeps the file
            in step with HEAD)."""
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            except (OSError, ValueError):
                return None, None
            key = "pdtv_half" if (kern == "pdtv" and args.half) else kern
            if key not in pmc or not default_shape or args.ring > 0.0 or getattr(args, "exact_tv", False):
                return None, None
            ent, cur = pmc[key], source_hash(kern)
            src = {"measured_on_sources": ent.get("sources_sha16"), "current_sources": cur, "profile": ent.get("profile")}
            return (ent["traffic_bytes"] if ent.get("sources_sha16") == cur else None), src

        traffic, traffic_src = committed_traffic(dom)
        # ... and measured live by this very run, as the cross-check of the committed figure (`traffic_committed` stays in
        # the line): two child rocprofv3 --pmc passes, by default only for the default workload on one GPU, where they are
        # known to fit beside this process (~25 s); elsewhere on request (--live-pmc); never with --no-pmc
        traffic_committed = traffic
        want_live = world == 1 and not args.no_pmc and (args.live_pmc or (default_shape and not args.overridden))
        if want_live:
            torch.cuda.empty_cache()
            live, how = live_traffic(dom, args, n, nz, sub)
            if live is not None:
                traffic = live
            traffic_src = dict(traffic_src or {}, live=how)
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": kernels[dom]["frac_hbm"], "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": kernels[dom]["avg_ms"], "launches": kernels[dom]["launches"],
                "bytes_per_launch": kernels[dom]["compulsory_bytes_per_launch"],
                "bytes_definition": "compulsory HBM bytes of one launch as executed (SURVEY 8d per-iteration bytes; a "
                                    "k-iteration fused PD_TV launch moves them once, not k times)"}
        if dom == "pdtv":
            roof["iterations_per_launch"] = kernels[dom]["iterations_per_launch"]
            roof["alg_GBps_per_iteration_equiv"] = kernels[dom]["alg_GBps_per_iteration_equiv"]
        if traffic_committed is not None:
            roof["traffic_committed"] = traffic_committed
        if traffic is not None:
            roof["traffic_GBps"] = traffic / kernels[dom]["avg_ms"] / 1e6
            roof["frac_traffic"] = roof["traffic_GBps"] / HBM_PEAK_GBS
        # BASELINE's metric names the back projector ("achieved HBM GB/s (backproj)"): its figures at the top level, whatever
        # kernel dominates the step.  `achieved` = SURVEY 8d's algorithmic bytes 4 (S_s + V) per call / the measured call; the
        # kernel is bound by LDS / VALU issue, not by HBM (docs/kernels/bp.md), so the LDS fractions ride along.
        roof_bp = None
        if "bp" in kernels:
            kb = kernels["bp"]
            bp_tr, bp_src = committed_traffic("bp")
            roof_bp = {"bound": "hbm", "kernel": "bp", "achieved": kb["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": kb["frac_hbm"], "traffic": bp_tr, "traffic_source": bp_src,
                       "avg_launch_ms": kb["avg_ms"], "launches": kb["launches"],
                       "bytes_per_launch": kb["compulsory_bytes_per_launch"], "updates_per_s": kb["updates_per_s"],
                       "frac_lds_spec_2.4GHz": kb["frac_lds"], "frac_lds_sustained_2.03GHz": kb["frac_lds_sustained"],
                       "note": "north-star kernel; an LDS-gather kernel: 8 B of ds_read_b128 per voxel-angle update put its floor at "
                               "lds_floor_ms, 4-5x above its HBM time (docs/kernels/bp.md)"}
        # ... and the forward projector's, the kernel that dominates configs[3] (826 of 1641 ms per ADMM iteration on one GPU)
        roof_fp = None
        if "fp" in kernels:
            kf = kernels["fp"]
            fp_tr, fp_src = committed_traffic("fp")
            roof_fp = {"bound": "hbm", "kernel": "fp", "achieved": kf["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": kf["frac_hbm"], "traffic": fp_tr, "traffic_source": fp_src,
                       "avg_launch_ms": kf["avg_ms"], "launches": kf["launches"],
                       "bytes_per_launch": kf["compulsory_bytes_per_launch"], "updates_per_s": kf["updates_per_s"],
                       "frac_lds_spec_2.4GHz": kf["frac_lds"], "frac_lds_sustained_2.03GHz": kf["frac_lds_sustained"],
                       "note": "one forward projection = its stepping-class launches; an LDS-gather kernel like the back projector "
                               "(8 B of ds_read_b128 per ray step), co-bound by VALU issue, staging 20-25 % of the call (docs/kernels/fp.md)"}
        units = args.steps if args.strong else args.steps * world
        what = (f"{nz_total} slices of {n}^2 split over {world} z-slab(s)" if args.strong
                else f"{nz} slices of {n}^2 per GPU; slab-iterations/s over {world} z-slab(s)")
        ring = f"+GH ring term (lambda {args.ring:g})" if args.ring > 0.0 else ""
        regs = "no regulariser" if args.reg == "none" else f"{args.reg}({args.inner} inner, {'f16' if args.half else 'f32'} duals)"
        loop = f"{args.method}-OS({args.os} subsets)" if args.os > 1 else args.method
        tag = f"BASELINE {args.baseline}" if not args.overridden else \
            f"custom: BASELINE {args.baseline} with {', '.join(args.overridden)} overridden"
        line = {
            "metric": "fista_os_iterations_per_sec", "value": units / dt, "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{loop}+{regs}{ring}, {na} angles, {what} ({tag})",
                       "slices_per_gpu": nz, "n": n, "angles": na, "os_number": args.os, "inner_iterations": args.inner,
                       "method": args.method, "slice_iterations_per_sec": args.steps * nz_total / dt,
                       "lipschitz_const": lc, "output_finite": finite,
                       "input": "analytic ellipsoid line integrals + Poisson noise (I0 2e4, peak attenuation 3)",
                       "backend": backend, "oversubscribed": oversubscribed,
                       **({"backend_note": backend_note} if backend_note else {})},
            "roofline": roof, "kernels": kernels,
        }
        if roof_bp is not None:
            line["roofline_bp"] = roof_bp
        if roof_fp is not None:
            line["roofline_fp"] = roof_fp
        if halo is not None:
            line["halo"] = halo
        # where the library put its TV scratch arena (docs/kernels/placement.md); "fast": False explains a PD_TV launch
        # that is 4-10 % slower than the same sources on a block of the fast class
        placed = _ops.placement_last()
        if placed is not None:
            line["placement"] = placed
        return line, sino, lc
    return None, sino, lc


def _lib_release(device):
    """Give the library's scratch arenas of this device back before a second workload is set up."""
    from tomobar_amd import _lib
    _lib.lib().tomo_release_scratch(int(device.index))


def north_star_child(args, world, ns_test, gpu_bytes):
    """Rank 0 only, after the headline: strong scaling of BASELINE configs[4] over `world` ranks as a child
    `torch.distributed.run` in its own process group, under a wall-clock limit.  Returns the `north_star` block."""
    import copy
    import signal
    from tomobar_amd.slab import slab_bounds
    block = {"workload": "BASELINE configs[4], --strong", "how": "supervised child job (second torch.distributed.run)"}
    ns = copy.copy(args)
    for k in ("n", "nz", "angles", "os", "inner", "reg", "method", "ring"):
        setattr(ns, k, None)
    ns.config, ns.strong, ns.half = "cfg5", True, False
    child_argv = ["--gpus", str(world), "--config", "cfg5", "--strong", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-pmc",
                  "--no-north-star", "--backend", args.backend]
    if ns_test:
        ns.n, ns.nz, ns.angles, ns.inner = 192, 12 * world, 96, 6
        child_argv += ["--n", "192", "--nz", str(12 * world), "--angles", "96", "--inner", "6"]
    apply_preset(ns)
    share = max(slab_bounds(ns.nz, world, r)[1] - slab_bounds(ns.nz, world, r)[0] for r in range(world))
    need = footprint_bytes(ns, share)
    block.update({"per_gpu_slices": share, "estimated_bytes_per_gpu": need})
    if need >= 0.9 * gpu_bytes:
        block["skipped"] = f"a {share}-slice slab needs ~{need / 1e9:.0f} GB of the GPU's {gpu_bytes / 1e9:.0f} GB"
        return block
    limit = float(os.environ.get("BENCH_NORTH_STAR_TIMEOUT", "900"))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                        "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TOMO_BENCH_ARGV", "BENCH_CONFIG", "BENCH_STRONG",
                        "BENCH_NORTH_STAR", "BENCH_NORTH_STAR_TEST")
           and not k.startswith(("TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC"))}
    env.update({"TOMO_BENCH_ARGV": json.dumps(child_argv), "TOMO_BENCH_CHILD": "1", "OMP_NUM_THREADS": env.get("OMP_NUM_THREADS", "1")})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
    if os.environ.get("BENCH_NORTH_STAR_CMD"):   # test seam (tests/test_host_logic.py): a stand-in child command, JSON list
        cmd = json.loads(os.environ["BENCH_NORTH_STAR_CMD"])
    def own_session_dies_with_parent():
        # its own session (= process group: one killpg reaches the launcher and every rank) and PR_SET_PDEATHSIG: should this
        # rank be killed by whoever supervises IT, the child launcher gets SIGTERM and takes its ranks down -- no orphaned job
        # keeps the GPUs of the next run busy
        os.setsid()
        try:
            C.CDLL(None).prctl(1, int(signal.SIGTERM))
        except Exception:  # noqa: BLE001
            pass

    t0 = time.perf_counter()
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                preexec_fn=own_session_dies_with_parent)
    except OSError as e:
        block["skipped"] = f"could not start the child job: {e!r}"[:300]
        return block
    try:
        out, err = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)      # the child's own session: its launcher and every rank, nothing else
        except OSError:
            pass
        out, err = proc.communicate()
        block["skipped"] = f"child job killed after the {limit:.0f} s wall-clock limit (BENCH_NORTH_STAR_TIMEOUT)"
        block["child_stderr_tail"] = (err or "")[-400:]
        return block
    block["child_wall_s"] = time.perf_counter() - t0
    child_line = None
    for ln in reversed((out or "").strip().splitlines()):
        try:
            cand = json.loads(ln)
        except ValueError:
            continue
        if isinstance(cand, dict) and "metric" in cand:
            child_line = cand
            break
    if proc.returncode != 0 or child_line is None:
        block["skipped"] = f"child job exited with code {proc.returncode}" + ("" if child_line is not None else " without a JSON line")
        block["child_stderr_tail"] = (err or "")[-400:]
        return block
    block.update({k: child_line[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "config",
                                             "roofline", "roofline_bp", "roofline_fp", "halo") if k in child_line})
    return block


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    # Several ranks: everything a library prints on stdout during the run (gloo announces its connections there: "[Gloo]
    # Rank 1 is connected to 3 peer ranks ...") goes to stderr, so that the ONE JSON line of rank 0 is all stdout carries.
    json_out = sys.stdout
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)

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
        dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
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

    env = dict(rank=rank, world=world, device=device, dev_index=dev_index, dist=dist, halo_group=halo_group,
               backend=backend, backend_note=backend_note, oversubscribed=oversubscribed)
    line, sino, lc = measure(args, env)
    if rank == 0 and not args.no_cpu and world == 1:
        line["cpu_baseline"] = cpu_baseline(args, sino, lc)
    del sino
    # N > 1: the headline value above is what the contract asks for (weak scaling of the default workload unless flags /
    # BENCH_CONFIG say otherwise).  The north-star target is STRONG scaling of BASELINE configs[4] (2560^2 x 2160, 1800
    # angles, OS 12, PD_TV + ring term).  It runs as a SUPERVISED CHILD JOB so that nothing it does can cost the headline:
    # every rank finishes the headline workload, the line is persisted on stderr, the ranks leave their process group and
    # give their GPU memory back; only then rank 0 starts a second `torch.distributed.run` of this script (--config cfg5
    # --strong) under a hard wall-clock limit, kills its whole process group if the limit passes, and merges the child's JSON
    # line into `north_star` -- or {"skipped": why} on a timeout / non-zero exit / a share that does not fit.
    ns_test = os.environ.get("BENCH_NORTH_STAR_TEST", "0") not in ("", "0")  # dry run of this block on a shared GPU, tiny shape
    ns_default = world >= 4 and not oversubscribed        # from 4 GPUs on every rank's share of configs[4] fits (<= 540 slices)
    want_ns = (world > 1 and (args.north_star or ns_default) and not args.no_north_star and (ns_test or not oversubscribed)
               and not (args.strong and args.config == "cfg5") and not os.environ.get("TOMO_BENCH_CHILD"))
    if want_ns:
        if rank == 0:
            print("[bench] headline (kept whatever the north-star child job does): " + json.dumps(line), file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        _lib_release(device)
        dist.barrier()                     # every rank is past its timed region AND has given its GPU memory back
        dist.destroy_process_group()
        dist = None
        if rank == 0:
            line["north_star"] = north_star_child(args, world, ns_test, torch.cuda.mem_get_info(device)[1])
    if rank == 0:
        print(json.dumps(line), file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
