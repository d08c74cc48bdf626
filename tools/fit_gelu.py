#!/usr/bin/env python3
"""Minimax fit of erf(x) = 1 - 2^(-p(x)), p(x) = x (c1 + c2 x + ... + c7 x^6) on [0, 4.3] (the GEGLU epilogue's erf, csrc/common.h
gelu_pair): weighted least squares on -log2(erfc(x)) / x with iterative reweighting towards the maximum absolute error of erf,
then the error of the fp32 Horner evaluation.  Prints the coefficients used in common.h (degree 7)."""
import numpy as np
from scipy.special import erf, erfc

X = 4.3
x = np.linspace(1e-6, X, 200001)
target = -np.log2(erfc(x))
for deg in (6, 7, 8):
    w = np.log(2) * erfc(x) * x
    A = np.vander(x, deg, increasing=True)
    ww = w.copy()
    for _ in range(200):
        coef = np.linalg.lstsq(A * ww[:, None], (target / x) * ww, rcond=None)[0]
        e = np.abs((1 - np.exp2(-(A @ coef) * x)) - erf(x))
        ww = ww * (1 + 2 * e / e.max())
        ww /= ww.mean()
    xf, cf = x.astype(np.float32), coef.astype(np.float32)
    acc = np.full_like(xf, cf[-1])
    for c in cf[-2::-1]:
        acc = (acc * xf + c).astype(np.float32)
    ef = (np.float32(1) - np.exp2(-(acc * xf).astype(np.float32)).astype(np.float32)).astype(np.float32)
    print(deg, "max |erf error| fp64 %.3g fp32 %.3g" % (e.max(), np.abs(ef.astype(np.float64) - erf(x)).max()), [f"{c:.9g}" for c in coef])
