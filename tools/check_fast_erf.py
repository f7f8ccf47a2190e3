"""Accuracy of the rational erf used by the GELU epilogues (csrc/mtt_device.h fast_erf), evaluated in float32 with numpy."""
import math

import numpy as np
from scipy.special import erf

A = [-2.72614225801306e-10, 2.77068142495902e-08, -2.10102402082508e-06, -5.69250639462346e-05, -7.34990630326855e-04,
     -2.95459980854025e-03, -1.60960333262415e-02]
B = [-1.45660718464996e-05, -2.13374055278905e-04, -1.68282697438203e-03, -7.37332916720468e-03, -1.42647390514189e-02]


def fast_erf(x):
    x = np.clip(x, -4, 4).astype(np.float32)
    x2 = (x * x).astype(np.float32)
    p = np.float32(A[0])
    for c in A[1:]:
        p = (p * x2 + np.float32(c)).astype(np.float32)
    q = np.float32(B[0])
    for c in B[1:]:
        q = (q * x2 + np.float32(c)).astype(np.float32)
    return (x * p / q).astype(np.float32)


if __name__ == "__main__":
    x = np.linspace(-8, 8, 4000001).astype(np.float32)
    e = np.abs(fast_erf(x) - erf(x.astype(np.float64))).max()
    g = np.abs(0.5 * x * (1 + fast_erf(x / np.float32(math.sqrt(2)))) - 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / math.sqrt(2)))).max()
    print(f"max |erf error| = {e:.3e}   max |gelu error| = {g:.3e}")
    assert e < 5e-7 and g < 2e-6
