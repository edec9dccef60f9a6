"""CPU oracle for the ToMoBAR FISTA/ADMM hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package ``tomobar_amd`` never does.

Kernels (FP / BP / PD-TV / ROF-TV) are the plain-C restatement in ``tomo_oracle.c``;
the outer loops below restate, in numpy float32 arithmetic, what the reference does in
``tomobar/methodsIR_CuPy.py`` (FISTA :401-484, ADMM :486-585, powermethod :311-354,
common initialisation :356-399), ``tomobar/data_fidelities.py:7-40`` and
``tomobar/regularisersCuPy.py`` (scalar set-up :215-218, 2D/3D squeeze :299-315).
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FLAG_LERP8 = 1


class Angle(C.Structure):
    _fields_ = [
        ("cs", C.c_float),
        ("sn", C.c_float),
        ("cor", C.c_float),
        ("slope", C.c_float),
        ("inv", C.c_float),
        ("scale", C.c_float),
        ("dirx", C.c_int32),
        ("src", C.c_int32),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libtomo_oracle.so")
    src = os.path.join(_HERE, "tomo_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libtomo_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        fp = C.POINTER(C.c_float)
        _LIB.orc_make_angles.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int,
                                         C.POINTER(C.c_int64), C.c_int, C.POINTER(Angle)]
        _LIB.orc_make_angles.restype = None
        for f in (_LIB.orc_fp3d, _LIB.orc_bp3d):
            f.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Angle), C.c_int]
            f.restype = None
        _LIB.orc_pdtv.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                  C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB.orc_pdtv.restype = C.c_int
        _LIB.orc_roftv.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                   C.c_int]
        _LIB.orc_roftv.restype = C.c_int
        _LIB.orc_round_half.argtypes = [C.c_float]
        _LIB.orc_round_half.restype = C.c_float
        _LIB.orc_set_threads.argtypes = [C.c_int]
        _LIB.orc_set_threads.restype = None
        _LIB.orc_max_threads.restype = C.c_int
        if "OMP_NUM_THREADS" not in os.environ:
            _LIB.orc_set_threads(usable_cpus())
    return _LIB


def usable_cpus() -> int:
    """CPUs this process can actually use: the affinity mask capped by the cgroup CPU quota (a container may show 256
    logical CPUs and grant 16 CPUs' worth of time; an OpenMP team of 256 on that quota is ~100x slower than one of 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = int(q) / int(p)
    except (OSError, ValueError):
        try:                                                               # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def threads() -> int:
    return int(lib().orc_max_threads())


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def os_indices(n_angles: int, os_number: int):
    """Interleaved subset table, astra_base.py:195-209, with the consumers' one-element trim
    (methodsIR_CuPy.py:454-456). Returns (newInd_Vec, NumbProjBins, list of trimmed index arrays)."""
    bins = int(np.ceil(float(n_angles) / float(os_number)))
    table = np.zeros((os_number, bins), dtype=np.int64)
    for s in range(os_number):
        for k in range(bins):
            idx = s + k * os_number
            if idx < n_angles:
                table[s, k] = idx
    subsets = []
    for s in range(os_number):
        v = table[s]
        if v[bins - 1] == 0:
            v = v[:-1]
        subsets.append(v.copy())
    return table, bins, subsets


class Projector:
    """Parallel-beam 3D operator pair on the geometry of supp/funcs.py:45-65."""

    def __init__(self, nz, n, nu, angles, cor=0.0, os_number=1, flags=0):
        self.nz, self.n, self.nu = int(nz), int(n), int(nu)
        self.angles = np.ascontiguousarray(angles, dtype=np.float64)
        self.na = self.angles.size
        self.vshift = None
        cor = np.asarray(cor, dtype=np.float64)
        if cor.ndim == 0:
            self.cor = cor.reshape(1).copy()
            self.cor_stride = 0
        elif cor.ndim == 1:
            self.cor = np.ascontiguousarray(cor)
            self.cor_stride = 1
        else:
            # [angles, 2] = (horizontal, vertical), supp/funcs.py:52-55.  The detector of angle a sits cor[a, 1] rows
            # higher, so its row r looks at slice r + cor[a, 1]: parallel rays stay in their slice and the operator is
            # the per-slice one composed with a per-angle linear resampling of the detector rows (shift_rows below).
            # Formula-level restatement: ASTRA, which would pin it, is not in this image ("parity unpinned" for this
            # component; the per-slice operators it wraps are pinned as before).
            self.cor = np.ascontiguousarray(cor)
            self.cor_stride = cor.shape[1]
            if np.any(cor[:, 1] != 0):
                self.vshift = np.ascontiguousarray(cor[:, 1], dtype=np.float32)
        self.flags = flags
        self.os_number = int(os_number) if os_number else 1
        self.full = self._table(None)
        if self.os_number > 1:
            self.newInd_Vec, self.NumbProjBins, self.subsets = os_indices(self.na, self.os_number)
            self.tables = [self._table(s) for s in self.subsets]
        else:
            self.subsets = [np.arange(self.na, dtype=np.int64)]
            self.tables = [self.full]

    def _table(self, index):
        nsel = self.na if index is None else len(index)
        tab = (Angle * max(nsel, 1))()
        idx_p = None
        if index is not None:
            index = np.ascontiguousarray(index, dtype=np.int64)
            idx_p = index.ctypes.data_as(C.POINTER(C.c_int64))
        lib().orc_make_angles(self.angles.ctypes.data_as(C.POINTER(C.c_double)),
                              self.cor.ctypes.data_as(C.POINTER(C.c_double)), self.cor_stride, self.na,
                              idx_p, nsel, tab)
        return tab, nsel

    def _sel(self, subset):
        return self.full if subset is None else self.tables[subset]

    def shift_rows(self, sino, subset, sign):
        """out[r, a, :] = (1 - w) sino[r + k, a, :] + w sino[r + k + 1, a, :], k + w = sign * vshift[a] (k = floor): the
        weight is the fractional part of the shift alone, the same for every row, so that a z-slab of the detector rows
        (tomobar_amd.slab) resamples exactly as the whole detector does; rows outside the detector read as zero.
        float32 throughout, one rounding per operation."""
        idx = np.arange(self.na) if subset is None else self.subsets[subset]
        sh = self.vshift[idx] * np.float32(sign)
        out = np.zeros_like(sino)
        r = np.arange(self.nz, dtype=np.int64)
        for a in range(sino.shape[1]):
            fl = np.floor(sh[a])
            w = np.float32(sh[a] - fl)
            r0 = r + int(fl)
            s0 = np.where(((r0 >= 0) & (r0 < self.nz))[:, None], sino[np.clip(r0, 0, self.nz - 1), a, :], np.float32(0))
            s1 = np.where(((r0 + 1 >= 0) & (r0 + 1 < self.nz))[:, None], sino[np.clip(r0 + 1, 0, self.nz - 1), a, :],
                          np.float32(0))
            out[:, a, :] = (np.float32(1) - w) * s0 + w * s1
        return out

    def fp(self, vol, subset=None):
        tab, nsel = self._sel(subset)
        vol = np.ascontiguousarray(vol, dtype=np.float32)
        assert vol.shape == (self.nz, self.n, self.n), vol.shape
        sino = np.empty((self.nz, nsel, self.nu), dtype=np.float32)
        lib().orc_fp3d(_fptr(vol), _fptr(sino), self.nz, self.n, self.nu, nsel, tab, self.flags)
        return sino if self.vshift is None else self.shift_rows(sino, subset, 1.0)

    def bp(self, sino, subset=None):
        tab, nsel = self._sel(subset)
        sino = np.ascontiguousarray(sino, dtype=np.float32)
        assert sino.shape == (self.nz, nsel, self.nu), (sino.shape, nsel)
        if self.vshift is not None:
            sino = np.ascontiguousarray(self.shift_rows(sino, subset, -1.0))
        vol = np.empty((self.nz, self.n, self.n), dtype=np.float32)
        lib().orc_bp3d(_fptr(sino), _fptr(vol), self.nz, self.n, self.nu, nsel, tab, self.flags)
        return vol


# ------------------------------------------------------------------ TV proximal operators
def _squeeze_2d(data):
    """regularisersCuPy.py:299-315"""
    if data.ndim == 2:
        return data, True, 0
    if data.ndim != 3:
        raise ValueError("2D or 3D arrays must be provided only")
    for i in range(3):
        if data.shape[i] == 1:
            return np.squeeze(data, axis=i), True, i
    return data, False, 0


def pd_scalars(regularisation_parameter, lipschitz_const):
    """regularisersCuPy.py:215-218 under NumPy-2 weak-scalar promotion (float32 arithmetic)."""
    tau = np.float32(regularisation_parameter * 0.1)
    sigma = np.float32(1.0 / (lipschitz_const * tau))
    theta = np.float32(1.0)
    lt = np.float32(tau / regularisation_parameter)
    return sigma, tau, lt, theta


def pd_tv(data, regularisation_parameter=1e-5, iterations=1000, methodTV=0, nonneg=0, lipschitz_const=8.0,
          half_precision=False):
    if data.dtype != np.float32:
        raise ValueError("The input data should be float32 data type")
    data, is2d, axis = _squeeze_2d(data)
    data = np.ascontiguousarray(data)
    sigma, tau, lt, theta = pd_scalars(regularisation_parameter, lipschitz_const)
    out = np.empty_like(data)
    if is2d:
        dy, dx = data.shape
        dz, nd = 1, 2
    else:
        dz, dy, dx = data.shape
        nd = 3
    rc = lib().orc_pdtv(_fptr(data), _fptr(out), dx, dy, dz, nd, sigma, tau, lt, theta, int(iterations),
                        int(bool(methodTV)), int(bool(nonneg)), int(bool(half_precision)))
    assert rc == 0
    return np.expand_dims(out, axis) if is2d else out


def rof_tv(data, regularisation_parameter=1e-5, iterations=3000, time_marching_parameter=0.001,
           half_precision=False):
    if data.dtype != np.float32:
        raise ValueError("The input data should be float32 data type")
    data, is2d, axis = _squeeze_2d(data)
    data = np.ascontiguousarray(data)
    out = np.empty_like(data)
    if is2d:
        dy, dx = data.shape
        dz, nd = 1, 2
    else:
        dz, dy, dx = data.shape
        nd = 3
    rc = lib().orc_roftv(_fptr(data), _fptr(out), dx, dy, dz, nd, np.float32(regularisation_parameter),
                         np.float32(time_marching_parameter), int(iterations), int(bool(half_precision)))
    if rc != 0:
        raise ValueError("ROF_TV needs every (squeezed) dimension >= 2")
    return np.expand_dims(out, axis) if is2d else out


def pd_step_slab(inp, u_in, u_out, p_in, p_out, dx, dy, nzl, has_lo, has_hi, sigma, tau, lt, theta, methodTV,
                 nonneg, half):
    """One PD_TV iteration on a ghosted z-slab (same signature as tomobar_amd.slab._hip_pd_step; CPU torch tensors).
    Used by the gloo tests of the halo-exchange logic."""
    import torch
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_pdtv_step.argtypes = [fp] * 9 + [C.c_int] * 7 + [C.c_float] * 4 + [C.c_int] * 3
    L.orc_pdtv_step.restype = None
    lo = 1 if has_lo else 0
    planes = nzl + lo + (1 if has_hi else 0)
    pin = [np.ascontiguousarray(p.numpy().astype(np.float32)) for p in p_in]
    pout = [np.zeros_like(a) for a in pin]
    uo = u_out.numpy()
    L.orc_pdtv_step(_fptr(inp.numpy()), _fptr(u_in.numpy()), _fptr(uo), _fptr(pin[0]), _fptr(pin[1]), _fptr(pin[2]),
                    _fptr(pout[0]), _fptr(pout[1]), _fptr(pout[2]), dx, dy, planes, lo, lo + nzl,
                    0 if has_lo else 1, 0 if has_hi else 1, sigma, tau, lt, theta, int(bool(methodTV)),
                    int(bool(nonneg)), int(bool(half)))
    for c in range(3):
        p_out[c][lo:lo + nzl] = torch.from_numpy(pout[c][lo:lo + nzl]).to(p_out[c].dtype)


def pd_pair_slab(inp, u_in, u_out, p_in, p_out, dx, dy, nzl, lo, hi, sigma, tau, lt, theta, methodTV, nonneg, half,
                 zr=None, k=2):
    """k (2 or 3) PD_TV iterations on a slab whose arrays carry lo / hi ghost planes (0 or >= k) (signature of
    tomobar_amd.slab._hip_pd_pair), as k applications of orc_pdtv_step: every application but the last also produces the
    ghost-zone planes the next one reads, so its output range shrinks by one plane per step towards the local planes."""
    import torch
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_pdtv_step.argtypes = [fp] * 9 + [C.c_int] * 7 + [C.c_float] * 4 + [C.c_int] * 3
    L.orc_pdtv_step.restype = None
    planes = nzl + lo + hi
    first_edge, last_edge = (0 if lo else 1), (0 if hi else 1)
    # stale garbage in never-consumed ghost planes must not be NaN for the CPU run: work on sanitised copies
    i_np = np.nan_to_num(inp.numpy().copy())
    u_cur = np.nan_to_num(u_in.numpy().copy())
    p_cur = [np.nan_to_num(np.ascontiguousarray(p.numpy().astype(np.float32))) for p in p_in]
    for j in range(k):
        shrink = k - 1 - j
        b = lo - shrink if lo else 0
        e = lo + nzl + shrink if hi else planes
        u_nxt = np.zeros_like(u_cur)
        p_nxt = [np.zeros_like(a) for a in p_cur]
        L.orc_pdtv_step(_fptr(i_np), _fptr(u_cur), _fptr(u_nxt), _fptr(p_cur[0]), _fptr(p_cur[1]), _fptr(p_cur[2]),
                        _fptr(p_nxt[0]), _fptr(p_nxt[1]), _fptr(p_nxt[2]), dx, dy, planes, b, e, first_edge, last_edge,
                        sigma, tau, lt, theta, int(bool(methodTV)), int(bool(nonneg)), int(bool(half)))
        u_cur, p_cur = u_nxt, p_nxt
    z0, z1 = zr if zr is not None else (0, nzl)  # only these local planes are written (tomo_pdtv_multi_slab_range)
    u_out[lo + z0:lo + z1] = torch.from_numpy(u_cur[lo + z0:lo + z1])
    for c in range(3):
        p_out[c][lo + z0:lo + z1] = torch.from_numpy(p_cur[c][lo + z0:lo + z1]).to(p_out[c].dtype)


def rof_step_slab(inp, u_in, u_out, dx, dy, nzl, lo, hi, lam, tau, half, zr=None):
    """One ROF_TV iteration on a ghosted z-slab (signature of tomobar_amd.slab._hip_rof_step); ``zr`` restricts the
    written local planes (tomo_roftv_iter_slab_range)."""
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_roftv_step.argtypes = [fp] * 3 + [C.c_int] * 7 + [C.c_float] * 2 + [C.c_int]
    L.orc_roftv_step.restype = None
    z0, z1 = zr if zr is not None else (0, nzl)
    L.orc_roftv_step(_fptr(inp.numpy()), _fptr(u_in.numpy()), _fptr(u_out.numpy()), dx, dy, nzl + lo + hi, lo + z0,
                     lo + z1, 0 if lo else 1, 0 if hi else 1, lam, tau, int(bool(half)))


def prox(X, reg, nonneg_regul):
    """regularisersCuPy.py:6-38"""
    if "ROF_TV" in reg["method"]:
        return rof_tv(X, reg["regul_param"], reg["iterations"], reg["time_marching_step"],
                      reg.get("half_precision", False))
    if "PD_TV" in reg["method"]:
        return pd_tv(X, reg["regul_param"], reg["iterations"], reg["methodTV"], nonneg_regul,
                     reg["PD_LipschitzConstant"], reg.get("half_precision", False))
    raise ValueError(reg["method"])


# ------------------------------------------------------------------ outer loops
def power_method(P: Projector, x1, iterations=15):
    """methodsIR_CuPy.py:323-354; x1 is the (normally random) start volume. OS uses subset 0 only."""
    sub = 0 if P.os_number > 1 else None
    x1 = np.asarray(x1, dtype=np.float32)
    y = P.fp(x1, sub)
    s = 1.0
    for _ in range(iterations):
        x1 = P.bp(y, sub)
        s = np.linalg.norm(np.ravel(x1))
        x1 = x1 / s
        y = P.fp(x1, sub)
    return float(s)


def pwls_weights(b):
    """methodsIR_CuPy.py:392-395"""
    w = np.maximum(np.asarray(b, dtype=np.float32), np.float32(1e-6))
    return w / w.max()


def grad_data_term(P, x, b_sub, subset, fidelity="LS", w_sub=None):
    """data_fidelities.py:28-40"""
    ax = P.fp(x, subset)
    if fidelity in ("LS", "PWLS"):
        res = ax - b_sub
        if w_sub is not None:
            res = res * w_sub
    elif fidelity == "KL":
        res = np.float32(1) - b_sub / np.clip(ax, np.float32(1e-8), None)
    else:
        raise ValueError(fidelity)
    return P.bp(res.astype(np.float32, copy=False), subset)


def robust_weight(res, huber=None, studentst=None):
    """Huber / Student's-t re-weighting of a data residual -- the data terms of the reference's removed RecToolsIR class
    (_data_["huber_threshold"], _data_["studentst_threshold"]: Demos/methods_IR_legacy/DemoFISTA_artifacts2D.py:197,263,
    307,348; listed in docs/source/introduction/about.rst:38).  NOT in this reference version: formula-level, PARITY UNPINNED.
    Huber: gradient of the Huber function, res * (delta/|res|) where |res| > delta.  Student's t [KAZ1_2017]: gradient of
    log(delta^2 + res^2), res * 2/(delta^2 + res^2).  float32 throughout."""
    res = np.asarray(res, dtype=np.float32)
    if huber is not None:
        d = np.float32(huber)
        mult = np.ones_like(res)
        big = np.abs(res) > d
        mult[big] = d / np.abs(res[big])
        res = mult * res
    if studentst is not None:
        d = np.float32(studentst)
        res = (np.float32(2.0) / (d * d + res * res)) * res
    return res.astype(np.float32)


def fista(P: Projector, b, iterations, lipschitz_const, nonnegativity=False, reg=None, fidelity="LS", x0=None,
          ring=None, beta_swls=0.1, huber=None, studentst=None):
    """methodsIR_CuPy.py:438-475 (b already padded, canonical [detY, angles, detX] layout).

    Ring-artefact data terms (NOT in this reference version -- supp/dicts.py:85-88 knows LS / PWLS / KL only; parity for
    them is formula-level, unpinned): documented for the reference's removed RecToolsIR class in
    docs/source/tutorials/real_data_recon.rst:100-151 with the models of docs/Kazantsev_CT_20.pdf Table III.
      * ``ring = {"lambda": l, "accelerate": c}``: Group-Huber, min_s 1/2||s - L^T r||^2 + l||s||_1 with
        L = (I (x) 1)/sqrt(m2), i.e. one offset per detector pixel [detY, detX], constant over the angles.  Per
        sub-iteration (the FISTA step on the offsets, as the removed class did it): res = (A_s x_t - b_s) + c * r_x;
        r = r_x - (1/L) * sum_angles(res) (float32, ascending angle order); PWLS weights are applied to res afterwards;
        after the image update r = soft(r, l) and r_x = r + ((t_old - 1)/t) (r - r_old).
      * ``fidelity = "SWLS"``: stripe-weighted least squares, W_s = W - W 1 (1^T W 1 + beta)^-1 1^T W per detector pixel:
        res_a = w_a res_a - w_a (sum_a w_a res_a)/(sum_a w_a + beta), sums over the sub-iteration's angles.
      * ``huber`` / ``studentst``: threshold of the robust re-weighting (robust_weight above) applied to the residual
        last, i.e. after the PWLS weights, the ring offsets and the SWLS weighting."""
    b = np.ascontiguousarray(b, dtype=np.float32)
    w = pwls_weights(b) if fidelity in ("PWLS", "SWLS") else None
    L_inv = np.float32(1.0 / lipschitz_const)
    X = np.zeros((P.nz, P.n, P.n), np.float32) if x0 is None else np.array(x0, dtype=np.float32)
    X_t = X.copy()
    t = np.float32(1.0)
    use_os = P.os_number > 1
    if ring is not None:
        r = np.zeros((P.nz, P.nu), np.float32)
        r_x = r.copy()
        lam, acc = np.float32(ring["lambda"]), np.float32(ring.get("accelerate", 50))
    for _ in range(iterations):
        for s in range(P.os_number):
            X_old, t_old = X, t
            sub = s if use_os else None
            idx = P.subsets[s] if use_os else slice(None)
            b_s = b[:, idx, :]
            w_s = None if w is None else w[:, idx, :]
            if ring is not None or fidelity == "SWLS":
                res = P.fp(X_t, sub) - b_s
                if ring is not None:
                    r_old = r
                    res = res + (acc * r_x)[:, None, :]
                    vec = np.zeros((P.nz, P.nu), np.float32)
                    for a in range(res.shape[1]):
                        vec = vec + res[:, a, :]
                    r = r_x - L_inv * vec
                    if fidelity == "PWLS":
                        res = res * w_s
                else:
                    wr = np.zeros((P.nz, P.nu), np.float32)
                    ws = np.zeros((P.nz, P.nu), np.float32)
                    for a in range(res.shape[1]):
                        wr = wr + w_s[:, a, :] * res[:, a, :]
                        ws = ws + w_s[:, a, :]
                    q = wr / (ws + np.float32(beta_swls))
                    res = w_s * res - w_s * q[:, None, :]
                if huber is not None or studentst is not None:
                    res = robust_weight(res, huber, studentst)
                grad = P.bp(np.ascontiguousarray(res, dtype=np.float32), sub)
            elif huber is not None or studentst is not None:
                # LS / PWLS residual of data_fidelities.py:28-34, re-weighted, then A^T
                res = P.fp(X_t, sub) - b_s
                if fidelity == "PWLS":
                    res = res * w_s
                grad = P.bp(np.ascontiguousarray(robust_weight(res, huber, studentst)), sub)
            else:
                grad = grad_data_term(P, X_t, b_s, sub, fidelity, w_s)
            X = X_t - L_inv * grad
            if nonnegativity:
                np.maximum(X, 0, out=X)
            if reg is not None and reg.get("method") is not None:
                X = prox(X, reg, 1 if nonnegativity else 0)
            t = np.float32((np.float32(1.0) + np.sqrt(np.float32(1.0) + np.float32(4.0) * t * t)) * np.float32(0.5))
            beta = np.float32((t_old - np.float32(1.0)) / t)
            X_t = X + beta * (X - X_old)
            if ring is not None:
                m = np.maximum(np.abs(r) - lam, np.float32(0.0))
                r = (np.sign(r) * m).astype(np.float32)
                r_x = r + beta * (r - r_old)
    return X


def admm(P: Projector, b, iterations, lipschitz_const, rho=1.0, relax=1.6, nonnegativity=False, reg=None,
         fidelity="LS", x0=None):
    """methodsIR_CuPy.py:515-566. ``reg['regul_param']`` is divided by rho on a copy."""
    b = np.ascontiguousarray(b, dtype=np.float32)
    w = pwls_weights(b) if fidelity == "PWLS" else None
    x = np.zeros((P.nz, P.n, P.n), np.float32) if x0 is None else np.array(x0, dtype=np.float32)
    z = x.copy()
    z_old = None
    u = np.zeros_like(x)
    tau = np.float32(0.9 / (lipschitz_const + rho))
    rho32 = np.float32(rho)
    if reg is not None and reg.get("method") is not None:
        reg = dict(reg)
        reg["regul_param"] = reg["regul_param"] / rho
    use_os = P.os_number > 1
    for it in range(iterations):
        for s in range(P.os_number):
            sub = s if use_os else None
            idx = P.subsets[s] if use_os else slice(None)
            b_s = b[:, idx, :]
            w_s = None if w is None else w[:, idx, :]
            grad = grad_data_term(P, z, b_s, sub, fidelity, w_s)
            grad_admm = rho32 * (z - x + u)
            z = z - tau * (grad + grad_admm)
            if nonnegativity:
                np.maximum(z, 0, out=z)
            if it > 1:
                z = np.float32(1.0 - relax) * z_old + np.float32(relax) * z
            z_old = z.copy()
            zu = z + u
            if reg is not None and reg.get("method") is not None:
                x = prox(zu, reg, 1 if nonnegativity else 0)
            else:
                x = zu
        u = u + (z - x)
    return x


def osem(P: Projector, b, iterations, nonnegativity=False, reg=None, x0=None):
    """methodsIR_CuPy.py:618-658 (OSEM; MLEM when os_number == 1): start from ones (dicts / __common_initialisation
    for method_run="OSEM"), ``normalisation = clip(A_0^T 1, 1e-8)`` from subset 0 only (:626-637), multiplicative update
    ``x *= A_s^T(b_s / clip(A_s x, 1e-8)) * normalisation`` (:648-654 -- the reference multiplies where the textbook
    form divides; restated as written), then the proximal step."""
    b = np.ascontiguousarray(b, dtype=np.float32)
    eps = np.float32(1e-8)
    use_os = P.os_number > 1
    x = np.ones((P.nz, P.n, P.n), np.float32) if x0 is None else np.array(x0, dtype=np.float32)
    sub0 = 0 if use_os else None
    n0 = len(P.subsets[0]) if use_os else P.na
    normalisation = np.clip(P.bp(np.ones((P.nz, n0, P.nu), np.float32), sub0), eps, None)
    for _ in range(iterations):
        for s in range(P.os_number):
            sub = s if use_os else None
            idx = P.subsets[s] if use_os else slice(None)
            ax = np.clip(P.fp(x, sub), eps, None)
            ratio = (b[:, idx, :] / ax).astype(np.float32, copy=False)
            back = P.bp(np.ascontiguousarray(ratio), sub)
            x = x * (back * normalisation)
            if reg is not None and reg.get("method") is not None:
                x = prox(x, reg, 1 if nonnegativity else 0)
    return x


# ------------------------------------------------------------------ FBP
def sinc_filter(n, cutoff, multiplier):
    """Half-spectrum, fftshift-ed sinc-ramp filter in float32: generate_filtersync.cu:5-82 / methodsDIR.py:295-312."""
    a = np.float32(cutoff)
    w = (np.float32(-np.pi) + np.arange(n, dtype=np.float32) * np.float32(2 * np.pi / n)).astype(np.float32)
    rd = (a * w / np.float32(2.0)).astype(np.float32)
    rn2 = np.sin(rd).astype(np.float32)
    dot = np.float32(np.sum((rn2 * rd / np.float32(np.sum(rd * rd, dtype=np.float32))).astype(np.float32), dtype=np.float32))
    r = (np.abs(np.float32(2.0) / a * rn2) * dot * dot).astype(np.float32)
    full = np.zeros(n, np.float32)
    full[(np.arange(n) + n // 2) % n] = r
    return (full[:n // 2 + 1] * np.float32(multiplier)).astype(np.float32)


def fbp_filter(data, cutoff=0.35):
    """fourier.py:26-78 on [angles, detY, detX] float32 data: rfft(detX) * filter, unnormalised irfft."""
    import scipy.fft
    na, nz, nu = data.shape
    f = sinc_filter(nu, cutoff, 1.0 / na / nu)
    spec = scipy.fft.rfft(data.astype(np.float32), axis=-1) * f
    return (scipy.fft.irfft(spec, nu, axis=-1) * nu).astype(np.float32)  # scipy normalises the inverse by 1/n


def fbp(P: Projector, data, cutoff=0.35):
    """methodsDIR_CuPy.py:114-150: filter, bring to [detY, angles, detX], back project."""
    return P.bp(np.ascontiguousarray(np.swapaxes(fbp_filter(data, cutoff), 0, 1)))


# ------------------------------------------------------------------ glue (suppTools.py)
def pad_detector(b, pad):
    """suppTools.py:425-459 (edge padding of detX)"""
    if pad <= 0:
        return b
    return np.pad(b, ((0, 0), (0, 0), (pad, pad)), mode="edge")


def crop_recon(vol, size):
    """suppTools.py:399-422"""
    n = vol.shape[2]
    a = (n - size) // 2
    return vol[:, a:a + size, a:a + size]


def circular_mask(vol, radius):
    """suppTools.py:364-396 (returns a masked copy)"""
    n = vol.shape[2]
    h = n // 2
    Y, X = np.ogrid[:n, :n]
    dist = np.sqrt((X - h) ** 2 + (Y - h) ** 2)
    if radius <= 1.0:
        mask = dist <= h - abs(h - h / radius)
    else:
        mask = dist <= h + abs(h - h / radius)
    return vol * mask


# ------------------------------------------------------------------ synthetic inputs
_SHEPP = [  # (A, a, b, c, x0, y0, z0, phi_deg) -- Kak-Slaney/Toft 3D head phantom, unit cube
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


def shepp_logan_3d(n, nz=None):
    """Voxelised ellipsoid phantom [nz, n, n] float32, unit cube mapped to n voxels."""
    nz = n if nz is None else nz
    xs = (np.arange(n) - n / 2 + 0.5) / (n / 2)
    zs = (np.arange(nz) - nz / 2 + 0.5) / (nz / 2)
    Z, Y, X = np.meshgrid(zs, xs, xs, indexing="ij")
    vol = np.zeros((nz, n, n), np.float32)
    for A, a, b, c, x0, y0, z0, phi in _SHEPP:
        p = np.deg2rad(phi)
        xr = (X - x0) * np.cos(p) + (Y - y0) * np.sin(p)
        yr = -(X - x0) * np.sin(p) + (Y - y0) * np.cos(p)
        vol += np.float32(A) * (((xr / a) ** 2 + (yr / b) ** 2 + ((Z - z0) / c) ** 2) <= 1.0)
    return vol


def shepp_logan_sino(n, nz, nu, angles):
    """Analytic line integrals of the same ellipsoids on the oracle's geometry: [nz, na, nu] float32
    (voxel units: lengths scaled by n/2)."""
    na = len(angles)
    sino = np.zeros((nz, na, nu), np.float64)
    s = (np.arange(nu) - nu / 2 + 0.5) / (n / 2)  # detector coordinate in unit-cube units
    zs = (np.arange(nz) - nz / 2 + 0.5) / (nz / 2)
    for A, a, b, c, x0, y0, z0, phi in _SHEPP:
        p = np.deg2rad(phi)
        for ia, th in enumerate(angles):
            # 2D ellipse of the z-section: semi-axes scale with sqrt(1 - ((z-z0)/c)^2)
            k2 = 1.0 - ((zs - z0) / c) ** 2  # [nz]
            valid = k2 > 0
            kk = np.sqrt(np.where(valid, k2, 0.0))
            al = th - p
            r2 = (a * np.cos(al)) ** 2 + (b * np.sin(al)) ** 2  # for unit section
            s0 = x0 * np.cos(th) + y0 * np.sin(th)
            d = (s[None, :] - s0)  # [1, nu]
            disc = r2 * (kk[:, None] ** 2) - d ** 2
            chord = np.where((disc > 0) & valid[:, None],
                             2.0 * a * b * np.sqrt(np.maximum(disc, 0.0)) / r2,
                             0.0)
            sino[:, ia, :] += A * chord
    return (sino * (n / 2)).astype(np.float32)
