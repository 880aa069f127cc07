"""oracle/resample.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

CPU restatement of the sample-rate conversion front end.  The reference has no
resampler (its CLI rejects a WAV whose rate differs from the model's, src/bin/birdnet-analyze.rs:447-455),
so there is nothing to pin against: the contract is the filter design written in include/birdnet_hip.h
(bn_recording_create_resampled), restated here independently in numpy float64:

  L/M = dst/src reduced;  fc = 0.5*min(1, L/M) cycles per source sample;  half = zc / (2 fc);
  T = 2*ceil(half) taps per phase;  h_p[j] = 2 fc sinc(2 fc tau) * kaiser_8.6(tau / half),
  tau = p/L - (j - (T/2 - 1)),  each phase divided by its sum;
  y[n] = sum_j h_{(nM) mod L}[j] * x[(nM) div L + j - (T/2 - 1)],  x = 0 outside the recording,
  int16 input first divided by 32768.

PARITY UNPINNED by the reference (no counterpart exists); cross-checked against scipy's resample_poly
on band-limited signals in the tests."""
from math import gcd

import numpy as np


def _i0(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    q = x * x / 4.0
    s = np.ones_like(x)
    term = np.ones_like(x)
    for k in range(1, 500):
        term = term * q / (k * k)
        s = s + term
        if np.all(term < 1e-18 * s):
            break
    return s


def make_table(src_rate: int, dst_rate: int, zero_crossings: int = 16):
    zc = zero_crossings or 16
    g = gcd(src_rate, dst_rate)
    L, M = dst_rate // g, src_rate // g
    fc = 0.5 * min(1.0, L / M)
    half = zc / (2.0 * fc)
    T = 2 * int(np.ceil(half))
    p = np.arange(L, dtype=np.float64)[:, None]
    j = np.arange(T, dtype=np.float64)[None, :]
    tau = p / L - (j - (T // 2 - 1))
    u = tau / half
    w = np.where(np.abs(u) < 1.0, _i0(8.6 * np.sqrt(np.clip(1.0 - u * u, 0.0, None))) / _i0(np.array(8.6)), 0.0)
    h = 2.0 * fc * np.sinc(2.0 * fc * tau) * w
    h = h / h.sum(axis=1, keepdims=True)
    return h.astype(np.float32), L, M, T


def resample(x: np.ndarray, src_rate: int, dst_rate: int, zero_crossings: int = 16) -> np.ndarray:
    x = np.asarray(x)
    xf = (x.astype(np.float32) / np.float32(32768.0)) if x.dtype == np.int16 else x.astype(np.float32)
    if src_rate == dst_rate:
        return xf
    table, L, M, T = make_table(src_rate, dst_rate, zero_crossings)
    n_src = len(xf)
    n_dst = (n_src * L + M - 1) // M
    pad = T
    xp = np.concatenate([np.zeros(pad, np.float64), xf.astype(np.float64), np.zeros(pad + 1, np.float64)])
    out = np.empty(n_dst, dtype=np.float32)
    CH = 1 << 16
    jj = np.arange(T)
    for s in range(0, n_dst, CH):
        n = np.arange(s, min(n_dst, s + CH), dtype=np.int64)
        pos = n * M
        base, phase = pos // L, pos % L
        idx = base[:, None] + jj[None, :] - (T // 2 - 1) + pad
        out[s:s + len(n)] = (table[phase].astype(np.float64) * xp[idx]).sum(axis=1).astype(np.float32)
    return out
