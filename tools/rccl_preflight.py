#!/usr/bin/env python3
"""10-second preflight of the multi-GPU transport on a node with several MI355X (round 4; nothing in this repository has
ever executed on more than one GPU -- this is the first thing to run there).

    python tools/rccl_preflight.py --gpus N          # self-launches N ranks, one per GPU, prints ONE JSON line

What it measures, per interior z-slab boundary (rank r <-> r+1), with the product's own exchange code
(tomobar_amd.slab.SlabComm: tomo_halo_pack -> one RCCL send + one recv per neighbour -> tomo_halo_unpack):
  * group creation: gloo default group + RCCL group beside it (what bench.py does), seconds each;
  * `exchange_ms`: one packed PD_TV halo of BASELINE configs[4] size -- 2560^2 planes, 12 planes up (U, P1..3 of 3 planes)
    and 9 down (U of 3, P1..3 of 2) = 550 MB per interior rank -- alone;
  * `kernel_ms`: one three-iteration PD_TV launch on a 128-slice slab of 2560^2 (every CU busy), alone;
  * `both_ms`: exchange_start -> the same launch -> exchange_wait, i.e. the overlap the slab driver assumes
    (DESIGN.md section 7).  `overlap` = (exchange_ms + kernel_ms - both_ms) / min(exchange_ms, kernel_ms): 1 = the
    transfer is hidden completely, 0 = it serialises with the kernel.
With fewer GPUs than ranks the ranks share devices and RCCL refuses (same device twice): the script then says so and
times the host-staged gloo path instead (`"backend": "gloo"`) -- a functional check, never a performance number.
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--n", type=int, default=2560, help="plane size (configs[4]: 2560)")
    ap.add_argument("--nz", type=int, default=128, help="slices of the slab the concurrent kernel works on")
    ap.add_argument("--reps", type=int, default=5)
    # ranks started below read their arguments from the environment: torch.distributed.run's own parser rejects options of
    # this script that are prefixes of its own (--n)
    args = ap.parse_args(json.loads(os.environ["TOMO_PREFLIGHT_ARGV"]) if "RANK" in os.environ and "TOMO_PREFLIGHT_ARGV" in os.environ
                         else None)
    if "RANK" not in os.environ:
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        env["TOMO_PREFLIGHT_ARGV"] = json.dumps(sys.argv[1:])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
        sys.exit(subprocess.call(cmd, env=env))

    # everything a library prints on stdout during the run (gloo announces its connections there) goes to stderr, so that
    # the ONE JSON line of rank 0 is all stdout carries
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from tomobar_amd.regularisersCuPy import PD_TV_cupy
    from tomobar_amd.slab import GHOST, SlabComm

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    oversub = world > ndev
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    t0 = time.perf_counter()
    dist.init_process_group("gloo")
    t_gloo = time.perf_counter() - t0
    group, backend, note, t_nccl = None, "gloo", None, None
    if not oversub:
        t0 = time.perf_counter()
        ok = 1
        try:
            group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
            x = torch.ones(8, device=dev)
            dist.all_reduce(x, group=group)
            torch.cuda.synchronize()
            ok = int(float(x[0].item()) == float(world))
        except Exception as e:  # noqa: BLE001
            ok, note = 0, repr(e)[:200]
        t_nccl = time.perf_counter() - t0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            backend = "nccl"
        else:
            group = None
    else:
        note = f"{world} ranks on {ndev} GPU(s): RCCL refuses two ranks on one device; host-staged gloo timed instead"
    comm = SlabComm(rank, world, dev, group=group)

    n = args.n
    plane = lambda k: torch.empty((k, n, n), dtype=torch.float32, device=dev)  # noqa: E731
    has_lo, has_hi = rank > 0, rank < world - 1
    send_up = [plane(GHOST) for _ in range(4)] if has_hi else []
    recv_up = [plane(GHOST)] + [plane(GHOST - 1) for _ in range(3)] if has_hi else []
    send_down = [plane(GHOST)] + [plane(GHOST - 1) for _ in range(3)] if has_lo else []
    recv_down = [plane(GHOST) for _ in range(4)] if has_lo else []
    for i, t in enumerate(send_up + send_down):
        t.fill_(float(rank * 10 + i))
    vol = torch.rand((args.nz, n, n), device=dev)
    out = torch.empty_like(vol)
    kernel = lambda: PD_TV_cupy(vol, 0.01, 3, 0, 1, 12.0, 0, False, out=out)  # noqa: E731

    def timed(fn):
        fn()
        torch.cuda.synchronize(); dist.barrier()
        ts = []
        for _ in range(args.reps):
            torch.cuda.synchronize(); dist.barrier()
            t = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]

    def exchange():
        comm.exchange(send_down, recv_down, send_up, recv_up)

    def both():
        h = comm.exchange_start(send_down, recv_down, send_up, recv_up)
        kernel()
        comm.exchange_wait(h)

    ex_ms, k_ms, b_ms = timed(exchange), timed(kernel), timed(both)
    # payload check: what arrived from below is what rank-1 sent up (its fill values)
    good = True
    if has_lo:
        good = all(float(t.flatten()[0].item()) == float((rank - 1) * 10 + i) for i, t in enumerate(recv_down))
    sent = sum(t.numel() * 4 for t in send_up + send_down)
    mine = dict(rank=rank, exchange_ms=ex_ms, kernel_ms=k_ms, both_ms=b_ms, bytes_sent=sent, payload_ok=bool(good),
                overlap=(ex_ms + k_ms - b_ms) / max(min(ex_ms, k_ms), 1e-9),
                GBps_per_direction=(sent / max(1, int(has_lo) + int(has_hi))) / ex_ms / 1e6 if sent else 0.0)
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank == 0:
        print(json.dumps({"tool": "rccl_preflight", "world": world, "gpus_visible": ndev, "backend": backend,
                          "note": note, "gloo_init_s": t_gloo, "rccl_group_s": t_nccl, "plane": [n, n],
                          "kernel": f"PD_TV x3 on {args.nz} x {n}^2", "ranks": allr}), file=json_out, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
