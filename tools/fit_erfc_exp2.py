#!/usr/bin/env python3
"""Fit behind the split-precision FFN block's GELU (csrc/ffn_block_f16x3.hip, round 5): erfc(z) = 2^(z q(z)) on [0, Z] with q a polynomial — one
v_exp_f32 per element, no reciprocal.  Reweighted least squares on Chebyshev nodes toward the minimax of the ERF error, then the
maximum error of a float32 Horner evaluation on a fine grid (printed per degree with the coefficients, lowest order first)."""
import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as C, polynomial as P
# GELU(a) = a * Phi(a), Phi(a) = 1 - 0.5 erfc(a / sqrt2) for a >= 0.  Fit q(z) = log2(erfc(z)) / z (p(0) = 0) on [0, Z], z = |a| / sqrt 2
Z = 4.2
def fit(deg):
    # Chebyshev nodes, iterative reweighting toward minimax of the ERF error: err_erf = erfc(z) * ln2 * z * dq
    n = 4000
    t = np.cos(np.pi * (np.arange(n) + 0.5) / n)
    z = (t + 1) * Z / 2
    z = np.maximum(z, 1e-9)
    target = np.log2(erfc(z)) / z
    w = erfc(z) * z
    coef = None
    wt = np.ones_like(z)
    for it in range(60):
        coef = C.chebfit(t, target, deg, w=w * wt)
        err = (C.chebval(t, coef) - target) * w * np.log(2)
        wt *= (1 + 3 * np.abs(err) / np.abs(err).max())
        wt /= wt.mean()
    # to monomial in z
    pc = C.cheb2poly(coef)            # in t
    # t = 2 z / Z - 1
    poly_t = np.polynomial.Polynomial(pc)
    poly_z = poly_t(np.polynomial.Polynomial([-1, 2 / Z]))
    return poly_z.coef
def eval32(c, z):
    z = z.astype(np.float32)
    acc = np.full_like(z, np.float32(c[-1]))
    for k in range(len(c) - 2, -1, -1):
        acc = acc * z + np.float32(c[k])        # (fma in hardware: slightly better)
    p = acc * z
    return np.float32(1) - np.exp2(p).astype(np.float32)
zz = np.linspace(0, 6, 2000001)
for deg in (5, 6, 7, 8, 9):
    c = fit(deg)
    e = eval32(c, np.minimum(zz, Z)).astype(np.float64)
    err = np.abs(e - erf(zz))
    print(deg, "max |erf err| float32 eval:", err.max(), "at z=", zz[err.argmax()], " coefs", [float(np.float32(x)) for x in c])
