#!/usr/bin/env python3
"""Fit the branch-free erf-GELU used in the GEMM epilogue:
       Phi(-a) = 0.5*erfc(a/sqrt2) = 2^-P(a),  a = min(|x|, AMAX);   gelu(x) = x * (x < 0 ? u : 1-u), u = 2^-P(|x|)
   P is a polynomial fitted (weighted by Phi, i.e. in absolute-Phi error) on [0, AMAX]; checked with an fp32 Horner emulation."""
import numpy as np
from scipy.special import erfc, erf

AMAX, DEG = 6.0, 11
f = lambda a: -np.log2(0.5 * erfc(a / np.sqrt(2)))
n = 6000
u = np.cos(np.pi * (np.arange(n) + 0.5) / n)
a = 0.5 * AMAX * (u + 1)
w = 0.5 * erfc(a / np.sqrt(2)) + 1e-4          # weight ~ Phi (+floor so the far tail stays sane)
V = np.vander(a / AMAX, DEG + 1, increasing=True)
coef, *_ = np.linalg.lstsq(V * w[:, None], f(a) * w, rcond=None)
coef = coef / AMAX ** np.arange(DEG + 1)        # polynomial in a
c32 = coef.astype(np.float32)

def horner32(c, x):
    r = np.full_like(x, c[-1], dtype=np.float32)
    for k in c[-2::-1]:
        r = (r.astype(np.float64) * x.astype(np.float64) + np.float64(k)).astype(np.float32)
    return r

def gelu32(x):
    x = x.astype(np.float32)
    aa = np.minimum(np.abs(x), np.float32(AMAX))
    uu = np.exp2(-horner32(c32, aa).astype(np.float32)).astype(np.float32)
    phi = np.where(x < 0, uu, (np.float32(1) - uu).astype(np.float32))
    return (x * phi).astype(np.float32)

xs = np.concatenate([np.linspace(-8, 8, 4000001), np.linspace(-1e-2, 1e-2, 20001)]).astype(np.float32)
ref = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
err = np.abs(gelu32(xs).astype(np.float64) - ref)
print("max abs gelu err", err.max(), "at", xs[err.argmax()])
print("max err / max(|gelu|,1e-2)", (err / np.maximum(np.abs(ref), 1e-2)).max())
print("coef:", ", ".join(f"{float(c):.9e}f" for c in c32))
