"""CPU restatement (numpy, float32 / complex64) of the reference's Fourier reconstruction on unequally spaced grids,
``RecToolsDIRCuPy.FOURIER_INV`` -- TEST INFRASTRUCTURE ONLY (imported by tests/ only; never by the product package).

Follows, stage by stage, /root/reference/tomobar/methodsDIR_CuPy.py:152-447 (driver), :449-545 (filter), :645-683
(pairing of slices into complex data), :701-836 (1D FFT + gathering), :851-897 (2D inverse FFT), :920-967 (unpadding),
the kernels of /root/reference/tomobar/cuda_kernels/fft_us_kernels.cu and the filter table of
/root/reference/tomobar/fourier.py:81-159.

Pinned: tests/golden/fourier_golden.npz holds reconstructions produced by the reference's own Python driver with its
own kernel source executed on the host (tests/golden/make_fourier_golden.py); tests/test_fourier.py checks this
restatement against all of them (<= 1e-5 relative L2).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.fft as sfft

F = np.float32
CENTER_SIZE_MIN = 192  # methodsDIR_CuPy.py:23
PI_F = F(3.1415926535897932384626433832795)  # fft_us_kernels.cu:2 (float literal)


# ----------------------------------------------------------------------------------------------- filter table
def wint(order: int, t: np.ndarray) -> np.ndarray:
    """Quadrature weights of fourier.py:81-108: the integral of s*p(s) over [t_j, t_{j+order-1}] for the polynomial
    p of degree order-1 through the samples, averaged over the windows that cover each node, linear tail of 40."""
    N = len(t)
    s = np.linspace(1e-40, 1, order)
    k = np.arange(order)
    vander = np.exp(np.outer(k, np.log(s)))                  # s^k, rows = powers
    inv_v = np.linalg.inv(vander)
    kk = np.arange(1, order + 2)
    prim = np.exp(np.outer(kk, np.log(s))) / kk[:, None]     # s^k / k
    u = np.diff(prim)                                        # integrals over the short intervals
    w1 = inv_v @ u[1:order + 1, :]
    w2 = inv_v @ u[0:order, :]
    overlap = 1.0 / np.concatenate((np.arange(1, order), (order - 1) * np.ones(N - 2 * (order - 1) - 1),
                                    np.arange(order - 1, 0, -1)))
    w = np.zeros(N)
    for j in range(N - order + 1):
        h = t[j + order - 1] - t[j]
        w[j:j + order] += (h * h * w1 + h * t[j] * w2) @ overlap[j:j + order - 1]
    w[-40:] = w[-40] / (N - 40) * np.arange(N - 40, N)
    return w


def calc_filter(ne: int, name: str, cutoff: float) -> np.ndarray:
    """fourier.py:111-159"""
    d = 0.5
    t = np.arange(0, ne / 2 + 1) / ne
    if name == "none":
        return np.asarray(ne * cutoff + t * 0, dtype=F)
    base = ne * cutoff * wint(12, t)
    window = {
        "ramp": lambda: 1.0,
        "shepp": lambda: np.sinc(t / (2 * d)) * (t / d <= 2),
        "cosine": lambda: np.cos(np.pi * t / (2 * d)) * (t / d <= 1),
        "cosine2": lambda: np.cos(np.pi * t / (2 * d)) ** 2 * (t / d <= 1),
        "hamming": lambda: (0.54 + 0.46 * np.cos(np.pi * t / d)) * (t / d <= 1),
        "hann": lambda: (1 + np.cos(np.pi * t / d)) / 2.0 * (t / d <= 1),
        "parzen": lambda: (1 - t / d) ** 3 * (t / d <= 1),
    }[name]()
    w = base * window
    w = 2 * w * (w >= 0)
    w[0] *= 2
    return np.asarray(w, dtype=F)


FILTERS = ("none", "ramp", "shepp", "cosine", "cosine2", "hamming", "hann", "parzen")


# ----------------------------------------------------------------------------------------------- stages
def oversampled_width(raw_n: int, n: int, pow2: bool = True, level: int = 4) -> int:
    """methodsDIR_CuPy.py:465-474"""
    if pow2:
        ne = 2 ** math.ceil(math.log2(raw_n * 3))
        if n > ne:
            ne = 2 ** math.ceil(math.log2(n))
        return ne
    return max(int(level * raw_n), n)


def fbp_filtering(data, raw_n, n, cor, filter_type, cutoff, pow2=True, level=4):
    """methodsDIR_CuPy.py:449-545: edge-pad to the oversampled width, rfft, filter x phase ramp, irfft, centre crop"""
    ne = oversampled_width(raw_n, n, pow2, level)
    pad_m = ne // 2 - raw_n // 2
    unpad_m, unpad_p = ne // 2 - n // 2, ne // 2 + n // 2
    t = sfft.rfftfreq(ne).astype(F)
    w = calc_filter(ne, filter_type, cutoff) * np.exp(-2 * np.pi * 1j * t * (cor + 0.5))
    w = w.astype(np.complex64)
    tmp = np.pad(data, ((0, 0), (0, 0), (pad_m, pad_m)), mode="edge")
    tmp = sfft.irfft(w * sfft.rfft(tmp, axis=2), axis=2)
    return np.ascontiguousarray(tmp[:, :, unpad_m:unpad_p]).astype(F)


def footprint_m(n: int, mu: float, eps: float) -> int:
    """methodsDIR_CuPy.py:726-737"""
    return int(np.ceil(2 * n * 1 / np.pi * np.sqrt(-mu * np.log(eps) + (mu * n) * (mu * n) / 4)))


def _clamp_half(v):
    return np.where(v >= F(0.5), F(0.5 - 1e-5), v).astype(F)


def gather_scatter(g, f, theta, m, mu, n, center_size=0, only_outside=False):
    """fft_us_kernels.cu:5-113 (gather_kernel / gather_kernel_partial): every polar sample adds its Gaussian footprint
    of (2m+1)^2 grid points (periodic wrap); with only_outside the grid points of the centre box are skipped."""
    nzh, nproj, _ = g.shape
    coeff0, coeff1 = PI_F / F(mu), -PI_F * PI_F / F(mu)
    chs = center_size // 2
    tx = np.arange(n)
    flat = f.reshape(nzh, -1)
    for ty in range(nproj):
        st, ct = F(np.sin(theta[ty])), F(np.cos(theta[ty]))
        r = ((tx - n // 2) / F(n)).astype(F)
        x0 = _clamp_half(r * ct)
        y0 = _clamp_half(-r * st)
        e0b = np.floor(F(2 * n) * x0).astype(np.int64) - m
        e1b = np.floor(F(2 * n) * y0).astype(np.int64) - m
        for i1 in range(2 * m + 1):
            ell1 = e1b + i1
            w1 = (ell1 / F(2 * n)).astype(F) - y0
            for i0 in range(2 * m + 1):
                ell0 = e0b + i0
                w0 = (ell0 / F(2 * n)).astype(F) - x0
                w = (coeff0 * np.exp(coeff1 * (w0 * w0 + w1 * w1).astype(F)).astype(F)).astype(F)
                keep = np.ones(n, bool)
                if only_outside:
                    keep = (ell0 < -chs) | (ell0 >= chs) | (ell1 < -chs) | (ell1 >= chs)
                ix = (ell0 + 3 * n) % (2 * n)
                iy = (ell1 + 3 * n) % (2 * n)
                idx = (ix + 2 * n * iy)[keep]
                vals = (g[:, ty, :][:, keep] * w[keep][None, :]).astype(np.complex64)
                np.add.at(flat, (slice(None), idx), vals)


def gather_center(g, f, theta, m, mu, n, center_size):
    """fft_us_kernels.cu:183-305 (angle pruning = the angles whose ray passes within the support radius of the grid
    point) + :376-517 (gather_kernel_center): every grid point of the centre box sums, over those angles in ascending
    angle order and over the radial samples inside the support circle, Gaussian-weighted samples."""
    nzh, nproj, _ = g.shape
    coeff0, coeff1 = PI_F / F(mu), -PI_F * PI_F / F(mu)
    chs = center_size // 2
    base = max(0, n - chs)
    tx = base + np.arange(center_size)
    TX, TY = np.meshgrid(tx, tx, indexing="xy")  # TX varies along columns
    px = ((TX - n).astype(F) / F(2 * n)).astype(F).ravel()
    py = ((n - TY).astype(F) / F(2 * n)).astype(F).ravel()
    radius_2 = F(2.0) * (F(m) + F(0.5)) * (F(m) + F(0.5)) / F(4 * n * n)
    acc = np.zeros((nzh, px.size), np.complex64)
    order = np.argsort(theta, kind="stable")
    pr, pr2 = F(0.5), F(0.25)
    for pi in order:
        st, ct = F(np.sin(theta[pi])), F(np.cos(theta[pi]))
        vx, vy = pr * ct, pr * st
        dot = (vx * px + vy * py).astype(F)
        mx, my = (dot * vx / pr2).astype(F), (dot * vy / pr2).astype(F)
        d2 = ((mx - px) * (mx - px) + (my - py) * (my - py)).astype(F)
        hit = radius_2 >= d2
        if not hit.any():
            continue
        dti = np.sqrt(np.maximum(radius_2 - d2, 0).astype(F)).astype(F)
        if abs(vx) > abs(vy):
            lo = np.floor(((mx - dti * vx / pr) / (F(2.0) * vx / F(n))).astype(F))
            hi = np.floor(((mx + dti * vx / pr) / (F(2.0) * vx / F(n))).astype(F))
        else:
            lo = np.floor(((my - dti * vy / pr) / (F(2.0) * vy / F(n))).astype(F))
            hi = np.floor(((my + dti * vy / pr) / (F(2.0) * vy / F(n))).astype(F))
        rmin = (n // 2 - 1 + lo).astype(np.int64)
        rmax = (n // 2 + 1 + hi).astype(np.int64)
        swap = rmin > rmax
        rmin, rmax = np.where(swap, rmax, rmin), np.where(swap, rmin, rmax)
        rmin = np.clip(rmin, 0, n - 1)
        rmax = np.clip(rmax, 0, n - 1)
        sel = np.nonzero(hit)[0]
        rmin_s, rmax_s, pxs, pys = rmin[sel], rmax[sel], px[sel], py[sel]
        span = int((rmax_s - rmin_s).max()) if sel.size else 0
        for k in range(span):
            ri = rmin_s + k
            ok = ri < rmax_s
            if not ok.any():
                break
            rr = ((ri - n // 2) / F(n)).astype(F)
            x0 = _clamp_half(rr * ct)
            y0 = _clamp_half(rr * st)
            w0, w1 = pxs - x0, pys - y0
            w = (coeff0 * np.exp((coeff1 * (w0 * w0 + w1 * w1).astype(F)).astype(F)).astype(F)).astype(F)
            vals = g[:, pi, np.clip(ri, 0, n - 1)] * w[None, :]
            acc[:, sel[ok]] += vals[:, ok].astype(np.complex64)
    f[:, base:base + center_size, base:base + center_size] = acc.reshape(nzh, center_size, center_size)


def fourier_inv(data, angles, cor, recon_size, detectors_x_pad=0, filter_type="shepp", cutoff_freq=1.0,
                center_size=32768, padding=0, power_of_2_oversampling=True, power_of_2_cropping=False,
                oversampling_level=4):
    """data: [detY, angles, detX] float32.  Returns the reconstruction [detY, recon, recon] (before the circular mask)."""
    data = np.asarray(data, dtype=F)
    nz, nproj, data_n = data.shape
    if recon_size > data_n:
        raise ValueError("The reconstruction size should not be larger than the size of the horizontal detector")
    odd_h, odd_v = data_n % 2, nz % 2
    data_n += odd_h
    nz += odd_v
    if odd_h or odd_v:  # :271-279
        p = np.zeros((nz, nproj, data_n), F)
        p[:nz - odd_v, :, :data_n - odd_h] = data
        if odd_h:
            p[:nz - odd_v, :, -1] = data[..., -1]
        data = p
    n = data_n + detectors_x_pad * 2 + padding * 2
    if power_of_2_cropping:
        n_pow2 = 2 ** math.ceil(math.log2(n))
        if 0.9 < n / n_pow2:
            n = n_pow2
    center_size = min(center_size, n * 2)
    theta = np.asarray(-np.asarray(angles), dtype=F)
    eps = 1e-4
    mu = -np.log(eps) / (2 * n * n)
    tmp_p = fbp_filtering(data, data_n, n, cor, filter_type, cutoff_freq, power_of_2_oversampling, oversampling_level)
    nzh = nz // 2
    sign = np.where(np.arange(n) % 2 == 1, F(1), F(-1)).astype(F)
    datac = ((tmp_p[0::2] + 1j * tmp_p[1::2]) * sign).astype(np.complex64)       # r2c_c1dfftshift
    datac = sfft.fft(datac, axis=-1).astype(np.complex64)
    m = footprint_m(n, mu, eps)
    datac = ((datac * F(4.0 / n)) * sign).astype(np.complex64)                   # c1dfftshift
    fde = np.zeros((nzh, 2 * n, 2 * n), np.complex64)
    if center_size >= CENTER_SIZE_MIN:
        if center_size != 2 * n:
            gather_scatter(datac, fde, theta, m, mu, n, center_size, only_outside=True)
        gather_center(datac, fde, theta, m, mu, n, center_size)
    else:
        gather_scatter(datac, fde, theta, m, mu, n)
    ii = np.arange(2 * n)
    chk = np.where((ii[None, :] % 2 == 0) != (ii[:, None] % 2 == 0), F(-1), F(1)).astype(F)  # c2dfftshift
    fde = (fde * chk).astype(np.complex64)
    fde = sfft.ifft2(fde, axes=(-2, -1)).astype(np.complex64)
    fde = (fde * chk).astype(np.complex64)
    # unpadding_mul_phi, :920-967 and fft_us_kernels.cu:605-658
    odd_r = recon_size % 2
    unpad_z = nz - odd_v
    um = (n - odd_h) // 2 - recon_size // 2
    up = (n - odd_h) // 2 + (recon_size + odd_r) // 2
    size = up - um
    r = um + np.arange(size)
    d = (F(-0.5) + (r * F(1.0) / F(n)).astype(F)).astype(F)
    phi = (np.exp((F(mu) * F(n * n) * (d[None, :] * d[None, :] + d[:, None] * d[:, None]).astype(F)).astype(F)).astype(F)
           * F(float(1 - n % 4) / nproj)).astype(F)
    blk = fde[:, n // 2 + um:n // 2 + up, n // 2 + um:n // 2 + up]
    rec = np.empty((2 * nzh, size, size), F)
    rec[0::2] = blk.real * phi
    rec[1::2] = blk.imag * phi
    return rec[:unpad_z]
