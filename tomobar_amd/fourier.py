"""Host-side tables of the Fourier reconstruction (``RecToolsDIRCuPy.FOURIER_INV``): the FBP filter family of the
reference (``tomobar/fourier.py:81-159``: ``none, ramp, shepp, cosine, cosine2, hamming, hann, parzen`` built on
12-point quadrature weights) and the size bookkeeping of ``tomobar/methodsDIR_CuPy.py:465-474,726-737``.  Small
float64 numpy work on the host, exactly as in the reference; everything per-sample runs in csrc/fourier_inv.hip."""
from __future__ import annotations

import math

import numpy as np

FILTER_NAMES = ("none", "ramp", "shepp", "cosine", "cosine2", "hamming", "hann", "parzen")
CENTER_SIZE_MIN = 192  # methodsDIR_CuPy.py:23


def _quadrature_weights(order: int, t: np.ndarray) -> np.ndarray:
    """Weights w_j such that sum_j w_j f(t_j) integrates s*f(s): on every window of ``order`` consecutive nodes the
    integrand is replaced by its interpolating polynomial, the contributions of the overlapping windows are averaged,
    and the last 40 weights follow a straight line (fourier.py:81-108)."""
    count = len(t)
    nodes = np.linspace(1e-40, 1, order)
    logs = np.log(nodes)
    powers = np.exp(np.outer(np.arange(order), logs))                 # nodes**k
    to_coeffs = np.linalg.inv(powers)
    k = np.arange(1, order + 2)
    primitives = np.exp(np.outer(k, logs)) / k[:, None]               # nodes**k / k
    pieces = np.diff(primitives)                                      # integrals over the short intervals
    lin_part = to_coeffs @ pieces[1:order + 1, :]
    const_part = to_coeffs @ pieces[0:order, :]
    share = 1.0 / np.concatenate((np.arange(1, order), np.full(count - 2 * (order - 1) - 1, order - 1.0),
                                  np.arange(order - 1, 0, -1)))
    w = np.zeros(count)
    for j in range(count - order + 1):
        h = t[j + order - 1] - t[j]
        w[j:j + order] += (h * h * lin_part + h * t[j] * const_part) @ share[j:j + order - 1]
    w[-40:] = w[-40] / (count - 40) * np.arange(count - 40, count)
    return w


def calc_filter(n: int, name: str, cutoff_freq: float) -> np.ndarray:
    """Half-spectrum filter of length n/2+1, float32 (fourier.py:111-159)."""
    if name not in FILTER_NAMES:
        raise ValueError(f"unknown filter {name!r}")
    d = 0.5
    t = np.arange(0, n / 2 + 1) / n
    if name == "none":
        return np.asarray(n * cutoff_freq + t * 0, dtype=np.float32)
    ramp = n * cutoff_freq * _quadrature_weights(12, t)
    if name == "ramp":
        w = ramp
    elif name == "shepp":
        w = ramp * np.sinc(t / (2 * d)) * (t / d <= 2)
    elif name == "cosine":
        w = ramp * np.cos(np.pi * t / (2 * d)) * (t / d <= 1)
    elif name == "cosine2":
        w = ramp * np.cos(np.pi * t / (2 * d)) ** 2 * (t / d <= 1)
    elif name == "hamming":
        w = ramp * (0.54 + 0.46 * np.cos(np.pi * t / d)) * (t / d <= 1)
    elif name == "hann":
        w = ramp * (1 + np.cos(np.pi * t / d)) / 2.0 * (t / d <= 1)
    else:  # parzen
        w = ramp * (1 - t / d) ** 3 * (t / d <= 1)
    w = 2 * w * (w >= 0)
    w[0] *= 2
    return np.asarray(w, dtype=np.float32)


def oversampled_width(raw_width: int, width: int, power_of_2: bool = True, level: int = 4) -> int:
    """methodsDIR_CuPy.py:465-474"""
    if power_of_2:
        ne = 2 ** math.ceil(math.log2(raw_width * 3))
        if width > ne:
            ne = 2 ** math.ceil(math.log2(width))
        return ne
    return max(int(level * raw_width), width)


def filter_with_phase(ne: int, name: str, cutoff_freq: float, rotation_axis: float) -> np.ndarray:
    """wfilter * exp(-2 pi i t rotation_axis), complex64 (methodsDIR_CuPy.py:479-483)"""
    t = np.fft.rfftfreq(ne).astype(np.float32)
    w = calc_filter(ne, name, cutoff_freq) * np.exp(-2 * np.pi * 1j * t * rotation_axis)
    return np.ascontiguousarray(w.astype(np.complex64))


def footprint_half_width(n: int, mu: float, eps: float) -> int:
    """methodsDIR_CuPy.py:726-737"""
    return int(np.ceil(2 * n * 1 / np.pi * np.sqrt(-mu * np.log(eps) + (mu * n) * (mu * n) / 4)))
