#!/usr/bin/env python3
"""The integer forms of CImg's linear resize in bgprep_fused_kernel (kernels.hip: fix_weight / enlarge_texel_fix, the
moving-average branch of cimg_resize_texel_pre) against the reference arithmetic they replace, on the CPU with numpy:
  enlarging   (unsigned char)((1 - a) v1 + a v2) in double  ==  v1 + ((d A1 + (d A0 >> 23)) >> 22)   for every weight with at
              most 45 fractional bits (all 256 x 256 byte pairs, thousands of weights incl. CImg's own running sums)
  shrinking   (unsigned char)(float(sum) / float(n))       ==  (sum * ceil(2^32 / n)) >> 32         at every sum next to a multiple of n
Also: weights that are NOT exact are detected as such."""
import numpy as np


def fix_weight(al):
    t = np.ldexp(al, 45)
    u = np.floor(np.ldexp(t, -23))
    return (t == np.floor(t)) & (al >= 0) & (al < 1), u.astype(np.int64), (t - np.ldexp(u, 23)).astype(np.int64)


def check_enlarging(rng):
    v1, v2 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    d = (v2 - v1).astype(np.int64)
    weights = [0.0, 0.5, 1 - 2.0 ** -45, 2.0 ** -45, 1 / 3 - (1 / 3) % 2.0 ** -45]
    weights += list(np.floor(rng.random(3000) * 2 ** 45) / 2 ** 45)               # 45 fractional bits
    weights += list(np.floor(rng.random(1000) * 2 ** 30) / 2 ** 30)               # fewer
    for n, s in ((853, 1024), (1023, 1024), (640, 768), (901, 2048), (200, 1024)):  # CImg's running sums, source positions >= 128
        f = (n - 1.0) / (s - 1)
        curr = 0.0
        for x in range(s):
            if curr >= 128:
                weights.append(curr - float(int(curr)))
            curr = min(n - 1.0, curr + f)
    n_exact = 0
    for al in weights:
        al = np.float64(al)
        ok, a1, a0 = fix_weight(al)
        if not ok:
            continue
        n_exact += 1
        ref = ((1 - al) * v1.astype(np.float64) + al * v2.astype(np.float64)).astype(np.uint8)  # C truncation of a value in [0, 256)
        p0 = d * a0
        assert np.abs(p0).max() < 2 ** 31 and np.abs(d * a1).max() < 2 ** 30
        got = v1 + ((d * a1 + (p0 >> 23)) >> 22)
        assert np.array_equal(got, ref), al
    assert n_exact > 4000
    # every running-sum weight at a source position >= 128 IS exact; below that some are not, and they are told apart
    for n, s in ((853, 1024), (700, 768)):
        f = (n - 1.0) / (s - 1)
        curr, inexact_low = 0.0, 0
        for x in range(s):
            ok, _, _ = fix_weight(np.float64(curr - float(int(curr))))
            if curr >= 128:
                assert ok, (n, s, x)
            elif not ok:
                inexact_low += 1
            curr = min(n - 1.0, curr + f)
        assert inexact_low > 0
    return n_exact


def check_shrinking():
    worst = 0
    for n in list(range(2, 2800)):
        m = 0xFFFFFFFF // n + 1
        k = np.arange(0, 256, dtype=np.int64)
        for off in (-1, 0, 1):
            acc = np.clip(k * n + off, 0, 255 * n)
            ref = (acc.astype(np.float32) / np.float32(n)).astype(np.uint8)  # correctly rounded float quotient, truncated
            got = (acc * m) >> 32
            assert np.array_equal(got, ref) and np.array_equal(got, acc // n), n
        worst = max(worst, 255 * n)
    assert worst < 2 ** 24
    return worst


if __name__ == "__main__":
    print("enlarging: %d exact weights x 65536 byte pairs agree" % check_enlarging(np.random.default_rng(1)))
    print("shrinking: sums up to %d agree with the float form" % check_shrinking())
