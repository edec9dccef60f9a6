"""z-slab sharding of the hot path across the GPUs of one node (no counterpart in the reference,
which scales by running independent replicas: Demos/methods_IR_legacy/MultiGPU_demo.py:144-190).

The parallel-beam projector pair is block-diagonal over z (detector row k <-> slice k), so every rank owns a contiguous
slab of slices of the volume(s) and the matching detector rows of the sinogram; forward / back projection and all
element-wise glue need no communication.  What does:

  * 3D TV couples neighbouring slices.  The slab arrays carry ghost planes which are refreshed by point-to-point
    send/recv between z-neighbours (RCCL over xGMI: one direct link per neighbour, no ring):
      - PD_TV runs up to THREE iterations per kernel pass (tomo_pdtv_multi_slab_range), so the ghosts are three planes
        deep and are refreshed once per launch: 12 planes up (U and P1..3 of the last three slices), 9 planes down (U
        of the first three slices, P1..3 of the first two), as 4 contiguous blocks per direction; a trailing single
        iteration uses tomo_pdtv_iter_slab on the same arrays;
      - ROF_TV: two planes of U up, one down, every iteration.
  * scalar reductions (power-method norm, PWLS weight maximum, CGLS inner products): all-reduce.

``SlabComm`` wraps ``torch.distributed`` (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests).  The TV drivers are
written against small "step" callables so that the same halo logic runs on the HIP kernels and, in the CPU tests, on
the oracle's single-iteration functions.
"""

from __future__ import annotations

import ctypes as C
import time
from typing import Callable, List, Optional

import numpy as np
import torch


def slab_bounds(nz_total: int, world: int, rank: int):
    """Contiguous slab [z0, z1) of rank `rank`, balanced: the first ``nz_total % world`` ranks hold one slice more
    than the others (floor / ceil split), so no rank is left short or empty while another holds a full share."""
    base, extra = divmod(int(nz_total), int(world))
    z0 = rank * base + min(rank, extra)
    return z0, z0 + base + (1 if rank < extra else 0)


def check_slab_split(nz_total: int, world: int, min_slices: int = 2):
    """Every rank evaluates this identical test BEFORE any exchange is posted, so a volume that is too short for the
    requested number of slabs raises on all ranks at once (a rank raising alone would leave its neighbours blocked in
    their send/recv).  3D TV needs at least two slices per slab (two-plane ghosts)."""
    if world > 1 and nz_total < min_slices * world:
        raise ValueError(f"{nz_total} slices cannot be split into {world} z-slabs of at least {min_slices} slices; "
                         f"use at most {max(nz_total // min_slices, 1)} ranks")


class _StagedRequest:
    """Completion handle of a host-staged transfer (gloo moving device tensors): wait() finishes the CPU transfer and,
    for a receive, copies the plane into the device tensor on the current stream."""

    def __init__(self, req, host, dev=None):
        self.req, self.host, self.dev = req, host, dev

    def wait(self):
        self.req.wait()
        if self.dev is not None:
            self.dev.copy_(self.host, non_blocking=False)


class SlabComm:
    """Neighbour exchange and scalar reductions for one rank of a z-slab decomposition.

    Backend "nccl" (RCCL over xGMI) moves device tensors directly.  With "gloo" device tensors are staged through host
    memory (functional path for the CPU tests and for several ranks sharing one GPU; not a performance path)."""

    def __init__(self, rank: int, world: int, device=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = int(rank), int(world)
        self.device = device
        self.group = group
        self.has_lo = self.rank > 0
        self.has_hi = self.rank < self.world - 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else "gloo"
        self.staged = self.backend != "nccl"   # tensors handed to the backend must live on the host
        self._stage_free = []   # packed-halo staging buffers not owned by an exchange in flight (see _take_staging)
        self._events = []
        self._wait_stream_ms = 0.0
        self.timing = False   # bench.py switches the HIP-event timing of the waits on
        self.stats = {"exchanges": 0, "messages": 0, "bytes": 0, "post_host_ms": 0.0, "wait_host_ms": 0.0}

    def _scalar_device(self):
        return None if self.staged else self.device

    # ---- scalars
    def allreduce_sum(self, value: float) -> float:
        t = torch.tensor([value], dtype=torch.float64, device=self._scalar_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def allreduce_max(self, value: float) -> float:
        t = torch.tensor([value], dtype=torch.float64, device=self._scalar_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def allgather_int(self, value: int):
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._scalar_device())
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        return [int(o.item()) for o in out]

    def barrier(self):
        self.dist.barrier(group=self.group)

    def validate_slabs(self, nz_local: int, min_slices: int = 2):
        """Collective check, run on EVERY call by EVERY rank (one int per rank): every rank learns every slab's height
        and ALL ranks raise together if one of them is too short for the ghost planes -- never a lone rank, which would
        strand its neighbours inside a send/recv.  (No per-rank cache: ranks whose local heights differ -- 9 slices over
        2 ranks is 5 + 4 -- would disagree about whether to join the all-gather.)"""
        if self.world == 1:
            return
        heights = self.allgather_int(nz_local)
        short = [r for r, h in enumerate(heights) if h < min_slices]
        if short:
            raise ValueError(f"z-slabs of ranks {short} hold fewer than {min_slices} slices (heights {heights}): "
                             f"3D TV needs {min_slices}-plane ghosts; use fewer ranks or balance the split with slab_bounds()")

    # ---- halo exchange
    def _take_staging(self, nbytes, device):
        """A staging buffer owned by ONE exchange from its post to the end of its wait(): exchange_start / exchange_wait
        allow several exchanges in flight, and a second post must neither pack into nor receive into a buffer an
        outstanding transfer still uses.  Buffers return to the pool in _Exchange.wait()."""
        nbytes = int(nbytes)
        best = None
        for i, buf in enumerate(self._stage_free):
            if buf.device == device and buf.numel() >= nbytes and (best is None or buf.numel() < self._stage_free[best].numel()):
                best = i
        buf = self._stage_free.pop(best) if best is not None else torch.empty(nbytes, dtype=torch.uint8, device=device)
        return buf

    def _post(self, send_down, recv_down, send_up, recv_up):
        """One exchange = at most ONE send and ONE receive per neighbour: the blocks of a direction (U, P1, P2, P3 planes,
        each contiguous in its own array) are packed into one staging buffer (tomo_halo_pack) and scattered back on
        arrival (tomo_halo_unpack).  Returns a handle whose wait() completes the transfers and the scatter."""
        t_host = time.perf_counter()
        P2POp = self.dist.P2POp
        plan, unpack, owned = [], [], []
        for name, tensors, peer, fn in (("sd", send_down, self.rank - 1, self.dist.isend),
                                        ("rd", recv_down, self.rank - 1, self.dist.irecv),
                                        ("su", send_up, self.rank + 1, self.dist.isend),
                                        ("ru", recv_up, self.rank + 1, self.dist.irecv)):
            if not tensors or peer < 0 or peer >= self.world:
                continue
            for t in tensors:
                if not t.is_contiguous():
                    raise ValueError("halo blocks must be contiguous plane ranges")
            nbytes = [t.numel() * t.element_size() for t in tensors]
            need = _halo_staging_bytes(nbytes)
            buf = self._take_staging(need, tensors[0].device)
            owned.append(buf)
            stage = buf[:need]
            if fn is self.dist.isend:
                _halo_pack(tensors, nbytes, stage)
            else:
                unpack.append((stage, tensors, nbytes))
            plan.append((fn, stage, peer))
            self.stats["messages"] += 1
            self.stats["bytes"] += int(sum(nbytes)) if fn is self.dist.isend else 0
        if not plan:
            return _Exchange(self, [], [], None, [])
        if not (self.staged and any(t.is_cuda for _, t, _ in plan)):
            reqs = list(self.dist.batch_isend_irecv([P2POp(fn, t, peer, self.group) for fn, t, peer in plan]))
        else:
            # host staging (gloo moving device tensors): the device->host copy of a packed buffer is synchronous with the
            # current stream, i.e. ordered after the pack kernel and the kernels that produced the planes
            ops, hosts = [], []
            for fn, t, peer in plan:
                h = t.cpu() if fn is self.dist.isend else torch.empty(t.shape, dtype=t.dtype)
                hosts.append((h, t if fn is self.dist.irecv else None))
                ops.append(P2POp(fn, h, peer, self.group))
            reqs = [_StagedRequest(r, h, d) for r, (h, d) in zip(self.dist.batch_isend_irecv(ops), hosts)]
        self.stats["exchanges"] += 1
        self.stats["post_host_ms"] += (time.perf_counter() - t_host) * 1e3
        return _Exchange(self, reqs, unpack, plan[0][1].device, owned)

    def exchange(self, send_down: List[torch.Tensor], recv_down: List[torch.Tensor],
                 send_up: List[torch.Tensor], recv_up: List[torch.Tensor]):
        """send_down/recv_down talk to rank-1, send_up/recv_up to rank+1.  All blocks are contiguous views; the k-th
        tensor sent up by rank r lands in the k-th tensor of rank r+1's recv_down (and likewise downwards)."""
        self._post(send_down, recv_down, send_up, recv_up).wait()

    def exchange_start(self, send_down, recv_down, send_up, recv_up):
        """Asynchronous form of `exchange`: returns the handle.  With the nccl (RCCL) backend the transfers are ordered
        after the work already queued on the current stream (the pack kernels included) and run on RCCL's own stream, so
        kernels launched after this call overlap with them; `exchange_wait` makes the current stream wait for their
        completion and scatters the received planes."""
        return self._post(send_down, recv_down, send_up, recv_up)

    @staticmethod
    def exchange_wait(handle):
        handle.wait()

    def timing_summary(self) -> dict:
        """Per-rank exchange statistics since construction (bench.py prints them): messages, payload bytes sent, host time
        spent posting (``post_host_ms``) and blocked in wait (``wait_host_ms``), and -- for device tensors -- the time the
        compute stream stood still between the start of a wait and the end of the scatter (``wait_stream_ms``, HIP events
        on the current stream; with RCCL this is what the exchange costs the kernels, the rest is overlapped)."""
        out = dict(self.stats)
        if self._events:
            torch.cuda.synchronize()
            self._wait_stream_ms += float(sum(a.elapsed_time(b) for a, b in self._events))
            self._events = []
        if self._wait_stream_ms or self.timing:
            out["wait_stream_ms"] = self._wait_stream_ms
        out["backend"] = self.backend
        return out


class _Exchange:
    """Handle of one packed halo exchange: wait() completes the requests and scatters what arrived."""

    def __init__(self, comm, reqs, unpack, device, owned):
        self.comm, self.reqs, self.unpack, self.device, self.owned = comm, reqs, unpack, device, owned

    def wait(self):
        comm = self.comm
        t_host = time.perf_counter()
        on_gpu = self.device is not None and self.device.type == "cuda" and comm.timing
        if on_gpu:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        for req in self.reqs:
            req.wait()
        for stage, tensors, nbytes in self.unpack:
            _halo_unpack(stage, tensors, nbytes)
        if on_gpu:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.device))
            comm._events.append((e0, e1))
        comm.stats["wait_host_ms"] += (time.perf_counter() - t_host) * 1e3
        # the staging buffers go back to the pool: on the GPU every later use is ordered after the scatter on the same stream
        comm._stage_free.extend(self.owned)
        self.reqs, self.unpack, self.owned = [], [], []


def check_ghost_rows(comm, g: int, rows: int, what: str):
    """Every rank raises, or none: some slab holds fewer than the ``g`` rows its neighbour needs as ghosts."""
    if comm.allreduce_max(1.0 if g > rows else 0.0) > 0.0:
        raise ValueError(f"{what} needs {g} ghost rows per slab boundary: some z-slab is thinner than that")


def extend_detector_rows(comm, sino: torch.Tensor, g: int):
    """The slab's projections ``[rows, angles, detX]`` with ``g`` ghost detector rows of each existing neighbour attached
    (one packed exchange): returns (extended array, index of the first local row in it).  Beyond the global first / last
    row nothing is attached -- a resampling of the extended rows reads zero there exactly as on the whole detector.
    Used by the vertical centre-of-rotation component, whose per-angle row resampling crosses slab boundaries
    (`HipTools3D._shift_rows`)."""
    rows = int(sino.shape[0])
    lo, hi = (g if comm.has_lo else 0), (g if comm.has_hi else 0)
    ext = torch.zeros((lo + rows + hi,) + tuple(sino.shape[1:]), dtype=sino.dtype, device=sino.device)
    ext[lo:lo + rows] = sino
    comm.exchange([sino[0:g]] if comm.has_lo else [], [ext[0:g]] if comm.has_lo else [],
                  [sino[rows - g:rows]] if comm.has_hi else [], [ext[lo + rows:]] if comm.has_hi else [])
    return ext, lo


def _halo_staging_bytes(nbytes):
    """tomo_halo_staging_bytes: every block starts 16-byte aligned in the staging buffer."""
    return sum((int(b) + 15) // 16 * 16 for b in nbytes)


def _halo_pack(tensors, nbytes, staging):
    """Device blocks: tomo_halo_pack (one kernel).  Host blocks (a SlabComm driven with CPU tensors over gloo, e.g. with
    custom step functions): plain slice copies -- the library only ever sees device pointers."""
    if staging.device.type != "cuda":
        off = 0
        for t, b in zip(tensors, nbytes):
            staging[off:off + int(b)].copy_(t.reshape(-1).view(torch.uint8))
            off += (int(b) + 15) // 16 * 16
        return
    _hip_halo_pack(tensors, nbytes, staging)


def _halo_unpack(staging, tensors, nbytes):
    if staging.device.type != "cuda":
        off = 0
        for t, b in zip(tensors, nbytes):
            t.reshape(-1).view(torch.uint8).copy_(staging[off:off + int(b)])
            off += (int(b) + 15) // 16 * 16
        return
    _hip_halo_unpack(staging, tensors, nbytes)


def _hip_halo_pack(tensors, nbytes, staging):
    from . import _lib as L
    from . import ops
    n = len(tensors)
    src = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nb = (C.c_size_t * n)(*[int(b) for b in nbytes])
    with torch.cuda.device(staging.device):
        L.check(L.lib().tomo_halo_pack(src, nb, n, ops.ptr(staging), ops.stream_ptr(staging)))


def _hip_halo_unpack(staging, tensors, nbytes):
    from . import _lib as L
    from . import ops
    n = len(tensors)
    dst = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nb = (C.c_size_t * n)(*[int(b) for b in nbytes])
    with torch.cuda.device(staging.device):
        L.check(L.lib().tomo_halo_unpack(ops.ptr(staging), dst, nb, n, ops.stream_ptr(staging)))


def _hip_pd_kmax(half: bool) -> int:
    """PD_TV iterations per fused launch, asked of the library (csrc/tv_kernels.hip: pd_iters_per_launch)."""
    if not torch.cuda.is_available():
        return 3   # host tensors (gloo functional path): the plan of the shipped kernels, three iterations per launch
    from . import _lib as L
    return int(L.lib().tomo_pdtv_iters_per_launch(int(bool(half))))


# ------------------------------------------------------------------------------------------------ PD_TV on a slab
GHOST = 3  # ghost planes below / above an interior boundary: as deep as the longest fused launch (3 iterations)


def pd_launch_plan(iterations: int, half: bool, kmax: Optional[int] = None):
    """How tomo_pdtv cuts `iterations` into fused launches (csrc/tv_kernels.hip: step_of): `kmax` iterations per launch
    (asked of the library: 3 in the shipped build, 2 under the exact-rounding test variant), as few single-iteration
    launches as possible (4 = 2 + 2).  The slab driver uses the same plan, so a slab run is launch for launch the
    whole-volume run."""
    if kmax is None:
        kmax = _hip_pd_kmax(half)
    kmax = max(1, min(int(kmax), GHOST))
    plan, rem = [], int(iterations)
    while rem > 0:
        if kmax >= 3 and rem >= 3 and rem != 4:
            k = 3
        else:
            k = 2 if (rem >= 2 and kmax >= 2) else 1
        plan.append(k)
        rem -= k
    return plan


class PdSlab:
    """Ghosted ping-pong state of a slab-sharded PD_TV run.

    Arrays address ``[lo + nz_local + hi][dy][dx]`` with ``lo = GHOST`` below an interior boundary (else 0) and
    ``hi = GHOST`` above one (else 0); the first local plane is index ``lo``."""

    def __init__(self, data: torch.Tensor, has_lo: bool, has_hi: bool, half: bool, pair_fn: Callable, step_fn: Callable,
                 alloc: Optional[Callable] = None):
        nzl, dy, dx = data.shape
        if (has_lo or has_hi) and nzl < GHOST:
            raise ValueError(f"PD_TV slabs must hold at least {GHOST} slices")
        self.nzl, self.dy, self.dx = nzl, dy, dx
        self.has_lo, self.has_hi = bool(has_lo), bool(has_hi)
        self.lo = GHOST if has_lo else 0
        self.hi = GHOST if has_hi else 0
        planes = nzl + self.lo + self.hi
        dev = data.device
        pd = torch.float16 if half else torch.float32
        self.half = bool(half)
        # only the initial duals need zeros; every other plane that influences an output is copied, received or
        # overwritten before it is read (the outermost Input ghost only feeds warm-up values that are discarded)
        # `alloc(specs, device)` (the HIP drivers pass ops.placed_empty): the nine work arrays as views of one block the
        # library places in HBM (docs/kernels/placement.md); default: nine allocations of the caller's allocator
        shape = (planes, dy, dx)
        specs = [(shape, torch.float32)] * 3 + [(shape, pd)] * 6
        self.lease = None
        if alloc is not None:
            arrs, self.lease = alloc(specs, dev)
        else:
            arrs = [torch.empty(sh, dtype=dt, device=dev) for sh, dt in specs]
        self.placed = alloc is not None   # the arrays belong to the library's block: results leave it as copies
        self.inp = arrs[0]
        self.inp[self.lo:self.lo + nzl] = data
        # the first step reads its primal variable from ``inp`` (U^0 = Input), so U[0] needs no initial copy
        self.U = arrs[1:3]
        self.first = True
        self.P = [arrs[3:6], arrs[6:9]]
        for t in self.P[0]:
            t.zero_()
        self.pair_fn, self.step_fn = pair_fn, step_fn
        self.cur = 0  # buffer set holding the current iterate

    def local(self, t: torch.Tensor) -> torch.Tensor:
        return t[self.lo:self.lo + self.nzl]

    def result(self) -> torch.Tensor:
        return self.local(self.inp if self.first else self.U[self.cur])

    def multi(self, k, sigma, tau, lt, theta, methodTV, nonneg):
        """k (2 or 3) iterations in one launch for every local plane."""
        self.multi_range(k, sigma, tau, lt, theta, methodTV, nonneg, 0, self.nzl)
        self.flip()

    def pair(self, sigma, tau, lt, theta, methodTV, nonneg):
        self.multi(2, sigma, tau, lt, theta, methodTV, nonneg)

    def multi_range(self, k, sigma, tau, lt, theta, methodTV, nonneg, z_begin, z_end):
        """k iterations for the local output planes [z_begin, z_end) only (buffer set cur -> cur ^ 1, no flip)."""
        i, o = self.cur, self.cur ^ 1
        if z_end > z_begin:
            self.pair_fn(self.inp, self._u_in(), self.U[o], self.P[i], self.P[o], self.dx, self.dy, self.nzl, self.lo,
                         self.hi, sigma, tau, lt, theta, methodTV, nonneg, self.half, (z_begin, z_end), k)

    def pair_range(self, sigma, tau, lt, theta, methodTV, nonneg, z_begin, z_end):
        self.multi_range(2, sigma, tau, lt, theta, methodTV, nonneg, z_begin, z_end)

    def _u_in(self) -> torch.Tensor:
        return self.inp if self.first else self.U[self.cur]

    def flip(self):
        self.cur ^= 1
        self.first = False

    def boundary_ranges(self):
        """Local plane ranges whose results the neighbours wait for (GHOST planes at an interior boundary) and the rest."""
        b0 = min(GHOST, self.nzl) if self.has_lo else 0
        b1 = max(self.nzl - (GHOST if self.has_hi else 0), b0)
        return ([(0, b0)] if self.has_lo else []) + ([(b1, self.nzl)] if self.has_hi and b1 < self.nzl else []), (b0, b1)

    def single(self, sigma, tau, lt, theta, methodTV, nonneg):
        """One iteration on the same arrays: the single-iteration kernel sees one ghost plane either side, i.e. the
        arrays shifted by GHOST - 1 planes where a ghost zone exists."""
        i, o = self.cur, self.cur ^ 1
        s = GHOST - 1 if self.has_lo else 0
        n = self.nzl + (1 if self.has_lo else 0) + (1 if self.has_hi else 0)
        v = lambda t: t[s:s + n]  # noqa: E731
        self.step_fn(v(self.inp), v(self._u_in()), v(self.U[o]), [v(p) for p in self.P[i]], [v(p) for p in self.P[o]],
                     self.dx, self.dy, self.nzl, self.has_lo, self.has_hi, sigma, tau, lt, theta, methodTV, nonneg,
                     self.half)
        self.flip()

    # ---- ghost planes of buffer set b.  Up = to rank+1 (its lo ghosts), down = to rank-1 (its hi ghosts).
    #      Up: U and P1..3 of the last GHOST planes; down: U of the first GHOST planes, P1..3 of the first GHOST - 1
    #      (the duals of the deepest upper ghost are never read).  Consecutive planes are one contiguous block, so an
    #      exchange is 4 messages per direction (U, P1, P2, P3), not one per plane.
    def send_up(self, b: int):
        if not self.has_hi:
            return []
        e = self.lo + self.nzl
        return [self.U[b][e - GHOST:e]] + [self.P[b][c][e - GHOST:e] for c in range(3)]

    def recv_down(self, b: int):
        if not self.has_lo:
            return []
        return [self.U[b][0:GHOST]] + [self.P[b][c][0:GHOST] for c in range(3)]

    def send_down(self, b: int):
        if not self.has_lo:
            return []
        f = self.lo
        return [self.U[b][f:f + GHOST]] + [self.P[b][c][f:f + GHOST - 1] for c in range(3)]

    def recv_up(self, b: int):
        if not self.has_hi:
            return []
        h = self.lo + self.nzl
        return [self.U[b][h:h + GHOST]] + [self.P[b][c][h:h + GHOST - 1] for c in range(3)]

    # ---- the one exchange before the first step: GHOST planes of Input either side (they are the ghosts of the initial
    #      primal variable as well, U^0 = Input); the initial duals are zero everywhere
    def initial_send_down(self):
        return [self.inp[self.lo:self.lo + GHOST]] if self.has_lo else []

    def initial_recv_down(self):
        return [self.inp[0:GHOST]] if self.has_lo else []

    def initial_send_up(self):
        e = self.lo + self.nzl
        return [self.inp[e - GHOST:e]] if self.has_hi else []

    def initial_recv_up(self):
        h = self.lo + self.nzl
        return [self.inp[h:h + GHOST]] if self.has_hi else []


PLACED_SLOT_PD, PLACED_SLOT_ROF = 0, 1   # one placed block per operator: a PD_TV and a ROF_TV solver on one stream never alias


def _hip_alloc(slot):
    def alloc(specs, device):
        from . import ops
        return ops.placed_empty(specs, device, slot=slot)
    return alloc


def _check_lease(st):
    """The work arrays are views of a library-owned block (ops.placed_empty): refuse to read a result out of a block that
    another solver on the same (device, stream, slot) has taken since -- e.g. two slab solvers interleaved on one stream."""
    if getattr(st, "lease", None) is not None:
        from . import ops
        if not ops.lease_is_current(st.lease):
            raise RuntimeError("the placed scratch block of this slab solver was handed to another solver on the same "
                               "(device, stream) before its result was read: run slab solvers one after the other per "
                               "stream, or on separate streams")


def _ptr3(ts):
    return (C.c_void_p * 3)(*[t.data_ptr() for t in ts])


def _hip_pd_step(inp, u_in, u_out, p_in, p_out, dx, dy, nzl, has_lo, has_hi, sigma, tau, lt, theta, methodTV, nonneg, half):
    from . import _lib as L
    from . import ops
    with torch.cuda.device(inp.device):
        L.check(L.lib().tomo_pdtv_iter_slab(inp.device.index, ops.ptr(inp), ops.ptr(u_in), ops.ptr(u_out), _ptr3(p_in),
                                            _ptr3(p_out), dx, dy, nzl, int(has_lo), int(has_hi), float(sigma),
                                            float(tau), float(lt), float(theta), int(bool(methodTV)), int(bool(nonneg)),
                                            int(bool(half)), ops.stream_ptr(inp)))


def _hip_pd_pair(inp, u_in, u_out, p_in, p_out, dx, dy, nzl, lo, hi, sigma, tau, lt, theta, methodTV, nonneg, half,
                 zr=None, k=2):
    from . import _lib as L
    from . import ops
    z0, z1 = zr if zr is not None else (0, nzl)
    with torch.cuda.device(inp.device):
        L.check(L.lib().tomo_pdtv_multi_slab_range(inp.device.index, ops.ptr(inp), ops.ptr(u_in), ops.ptr(u_out),
                                                   _ptr3(p_in), _ptr3(p_out), dx, dy, nzl, int(lo), int(hi), int(z0),
                                                   int(z1), int(k), float(sigma), float(tau), float(lt), float(theta),
                                                   int(bool(methodTV)), int(bool(nonneg)), int(bool(half)),
                                                   ops.stream_ptr(inp)))


def pd_tv_slab(data: torch.Tensor, comm, regularisation_parameter, iterations, methodTV=0, nonneg=0,
               lipschitz_const=8.0, half_precision=False, pair_fn: Optional[Callable] = None,
               step_fn: Optional[Callable] = None, out=None, overlap: bool = True):
    """PD_TV of a z-slab of a larger 3D volume; bit-identical to running PD_TV_cupy on the whole volume."""
    tau = np.float32(regularisation_parameter * 0.1)
    sigma = np.float32(1.0 / (lipschitz_const * tau))
    theta = np.float32(1.0)
    lt = np.float32(tau / regularisation_parameter)
    comm.validate_slabs(data.shape[0], GHOST)
    st = PdSlab(data, comm.has_lo, comm.has_hi, half_precision, pair_fn or _hip_pd_pair, step_fn or _hip_pd_step,
                alloc=_hip_alloc(PLACED_SLOT_PD) if (pair_fn is None and step_fn is None and data.is_cuda) else None)
    comm.exchange(st.initial_send_down(), st.initial_recv_down(), st.initial_send_up(), st.initial_recv_up())
    edge_ranges, interior = st.boundary_ranges()
    # Overlap: the planes the neighbours wait for are computed first (two thin launches), their exchange runs on RCCL's
    # stream while the interior of the slab is computed, and the next launch starts when both are done.
    overlap = overlap and bool(edge_ranges) and interior[1] - interior[0] >= 4
    plan = pd_launch_plan(iterations, half_precision)
    for n, k in enumerate(plan):
        more = n + 1 < len(plan)
        if k >= 2 and overlap and more:
            for z0, z1 in edge_ranges:
                st.multi_range(k, sigma, tau, lt, theta, methodTV, nonneg, z0, z1)
            b = st.cur ^ 1
            reqs = comm.exchange_start(st.send_down(b), st.recv_down(b), st.send_up(b), st.recv_up(b))
            st.multi_range(k, sigma, tau, lt, theta, methodTV, nonneg, interior[0], interior[1])
            st.flip()
            comm.exchange_wait(reqs)
            continue
        if k >= 2:
            st.multi(k, sigma, tau, lt, theta, methodTV, nonneg)
        else:
            st.single(sigma, tau, lt, theta, methodTV, nonneg)
        if more:
            b = st.cur
            comm.exchange(st.send_down(b), st.recv_down(b), st.send_up(b), st.recv_up(b))
    _check_lease(st)
    res = st.result()  # a view of the last output buffer (the buffer lives as long as the view)
    if iterations == 0:
        res = data
    if out is not None:
        out.copy_(res)
        return out
    return res.clone() if (st.placed and iterations > 0) else res


# ------------------------------------------------------------------------------------------------ ROF_TV on a slab
class RofSlab:
    """Ghosted ping-pong state for ROF_TV: two ghost planes of U below (D3 of the plane below needs U two planes
    down), one above."""

    def __init__(self, data: torch.Tensor, has_lo: bool, has_hi: bool, half: bool, step_fn: Callable,
                 alloc: Optional[Callable] = None):
        nzl, dy, dx = data.shape
        self.nzl, self.dy, self.dx = nzl, dy, dx
        self.lo = 2 if has_lo else 0
        self.hi = 1 if has_hi else 0
        planes = nzl + self.lo + self.hi
        dev = data.device
        self.half = bool(half)
        specs = [((planes, dy, dx), torch.float32)] * 3
        self.lease = None
        if alloc is not None:
            arrs, self.lease = alloc(specs, dev)
        else:
            arrs = [torch.empty(sh, dtype=dt, device=dev) for sh, dt in specs]
        self.inp = arrs[0]
        self.inp[self.lo:self.lo + nzl] = data
        self.U = arrs[1:3]
        self.U[0][self.lo:self.lo + nzl] = data
        self.step_fn = step_fn

    def local(self, t):
        return t[self.lo:self.lo + self.nzl]

    def step(self, it, lam, tau, zr=None):
        """iteration ``it``: buffer it&1 -> (it+1)&1, all local planes or only the local range ``zr``"""
        if zr is None:
            self.step_fn(self.inp, self.U[it & 1], self.U[(it + 1) & 1], self.dx, self.dy, self.nzl, self.lo, self.hi,
                         lam, tau, self.half)
        elif zr[1] > zr[0]:
            self.step_fn(self.inp, self.U[it & 1], self.U[(it + 1) & 1], self.dx, self.dy, self.nzl, self.lo, self.hi,
                         lam, tau, self.half, zr)

    def boundary_ranges(self):
        """planes the neighbours wait for (first plane below an interior boundary, last two above one) and the rest"""
        b0 = 1 if self.lo else 0
        b1 = self.nzl - (2 if self.hi else 0)
        return ([(0, b0)] if self.lo else []) + ([(b1, self.nzl)] if self.hi else []), (b0, b1)

    def send_down(self, b):
        return [self.U[b][self.lo]]

    def send_up(self, b):
        last = self.lo + self.nzl - 1
        return [self.U[b][last - 1:last + 1]]   # two consecutive planes: one contiguous block

    def recv_down(self, b):
        return [self.U[b][0:2]] if self.lo else []

    def recv_up(self, b):
        return [self.U[b][self.lo + self.nzl]] if self.hi else []


def _hip_rof_step(inp, u_in, u_out, dx, dy, nzl, lo, hi, lam, tau, half, zr=None):
    from . import _lib as L
    from . import ops
    z0, z1 = zr if zr is not None else (0, nzl)
    with torch.cuda.device(inp.device):
        L.check(L.lib().tomo_roftv_iter_slab_range(inp.device.index, ops.ptr(inp), ops.ptr(u_in), ops.ptr(u_out), dx, dy,
                                                   nzl, int(lo), int(hi), int(z0), int(z1), float(lam), float(tau),
                                                   int(bool(half)), ops.stream_ptr(inp)))


def rof_tv_slab(data: torch.Tensor, comm, regularisation_parameter, iterations, time_marching_parameter,
                half_precision=False, step_fn: Optional[Callable] = None, out=None, overlap: bool = True):
    comm.validate_slabs(data.shape[0])
    st = RofSlab(data, comm.has_lo, comm.has_hi, half_precision, step_fn or _hip_rof_step,
                 alloc=_hip_alloc(PLACED_SLOT_ROF) if (step_fn is None and data.is_cuda) else None)
    lam, tau = np.float32(regularisation_parameter), np.float32(time_marching_parameter)
    comm.exchange(st.send_down(0), st.recv_down(0), st.send_up(0), st.recv_up(0))
    edge_ranges, interior = st.boundary_ranges()
    overlap = overlap and bool(edge_ranges) and interior[1] - interior[0] >= 4
    for it in range(iterations):
        more = it + 1 < iterations
        b = (it + 1) & 1
        if overlap and more:  # boundary planes, exchange in flight, interior (see pd_tv_slab)
            for zr in edge_ranges:
                st.step(it, lam, tau, zr)
            reqs = comm.exchange_start(st.send_down(b), st.recv_down(b), st.send_up(b), st.recv_up(b))
            st.step(it, lam, tau, interior)
            comm.exchange_wait(reqs)
            continue
        st.step(it, lam, tau)
        if more:
            comm.exchange(st.send_down(b), st.recv_down(b), st.send_up(b), st.recv_up(b))
    _check_lease(st)
    res = st.local(st.U[iterations & 1])
    if out is not None:
        out.copy_(res)
        return out
    return res.clone()
