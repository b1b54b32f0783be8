#!/usr/bin/env python3
"""Derivation check for byte_over_255 (csrc/kernels.hip): for every byte u, is

    q0 = rn(u * c),  c = rn(1/255);   q = fma(fma(-q0, 255, u), c, q0)

the correctly rounded fp32 quotient u / 255.f the reference's composite masks divide by (DG:606, 626)?  Exact rational
arithmetic, each fma rounded once.  Prints the bytes where the bare product and the Newton step differ from the division."""
from fractions import Fraction

import numpy as np

f32 = np.float32


def rn(fr):
    """Fraction -> nearest float32 (candidates around the double-rounded value, compared exactly)."""
    x = f32(float(fr))
    cands = [np.nextafter(x, f32(-np.inf)), x, np.nextafter(x, f32(np.inf))]
    return min(cands, key=lambda c: abs(Fraction(float(c)) - fr))


def newton_wrong():
    c = f32(1) / f32(255)
    assert float(c).hex() == "0x1.0101020000000p-8"
    mul, newton = [], []
    for u in range(256):
        q = f32(u) / f32(255)
        q0 = f32(u) * c
        r = rn(Fraction(u) - Fraction(float(q0)) * 255)
        q1 = rn(Fraction(float(r)) * Fraction(float(c)) + Fraction(float(q0)))
        if q0 != q:
            mul.append(u)
        if q1 != q:
            newton.append(u)
    return mul, newton


if __name__ == "__main__":
    mul, newton = newton_wrong()
    print("u * rn(1/255) differs from u / 255.f for %d bytes; after one Newton step for %d" % (len(mul), len(newton)))
