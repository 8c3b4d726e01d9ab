"""CPU: the identity behind the line-search kernels' row-level harmonic sum (chicdiff_amd/csrc/disp_kernels.hip, round 6).

DESeq2's dlog_posterior (fitDisp, reached from chicdiff.R:1573 / 1602 / 1643 / 1673) needs, per sample,
digamma(y + r) - digamma(r).  The kernels split it as H_n + [digamma(y + r) - digamma(r + n)] with n = min(y, nr), nr = the unit
steps that lift r to >= 10 and H_n = sum_{i<n} 1 / (r + i).  Rounds 1-5 tabulated H_n per tick; round 6 adds the H_n of ALL samples at
row level from the row's count profile c_i = #{j : y_j > i}:

    sum_j H_{min(y_j, nr)} = sum_{i < nr} c_i / (r + i),      c_i ten bytes in three 32-bit words (profile_add / harmonic_row).

This file checks the identity, the packing, and the masking by nr in a numpy twin of the device code — no GPU, no oracle."""
import numpy as np
from scipy import special


def profile_words(y):
    """profile_add() over the samples of one row: bytes [0, min(y_j, 10)) of an 80-bit word get + 1 per sample."""
    w = [0, 0, 0]
    for yj in y:
        m = int(min(yj, 10))
        a, b, d = min(m, 4), (0 if m < 4 else min(m - 4, 4)), (0 if m < 8 else m - 8)
        w[0] += 0x01010101 & (0xFFFFFFFF if a == 4 else (1 << (8 * a)) - 1)
        w[1] += 0x01010101 & (0xFFFFFFFF if b == 4 else (1 << (8 * b)) - 1)
        w[2] += 0x00000101 & ((1 << (8 * d)) - 1)
    return w


def harmonic_row(w, r, nr):
    """harmonic_row<false>(): mask the profile to its first nr bytes, then ten fused steps in i order."""
    a, b, d = min(nr, 4), (0 if nr < 4 else min(nr - 4, 4)), (0 if nr < 8 else nr - 8)
    w0 = w[0] & (0xFFFFFFFF if a == 4 else (1 << (8 * a)) - 1)
    w1 = w[1] & (0xFFFFFFFF if b == 4 else (1 << (8 * b)) - 1)
    w2 = w[2] & ((1 << (8 * d)) - 1)
    h, zz = 0.0, r
    for i in range(10):
        word = w0 if i < 4 else (w1 if i < 8 else w2)
        h += float((word >> (8 * (i & 3))) & 0xFF) / zz
        zz += 1.0
    return h


def test_profile_bytes_are_the_counts_above_i():
    rng = np.random.default_rng(1)
    for S in (1, 3, 8, 16, 64):
        for _ in range(50):
            y = rng.choice([0, 0, 1, 2, 3, 5, 9, 10, 11, 40, 2**31 - 1], size=S)
            w = profile_words(y)
            for i in range(10):
                word = w[0] if i < 4 else (w[1] if i < 8 else w[2])
                assert (word >> (8 * (i & 3))) & 0xFF == int((y > i).sum()), (S, i)
            assert w[2] >> 16 == 0  # bytes 10, 11 stay empty


def test_row_level_harmonic_sum_equals_the_sum_of_the_samples_harmonic_sums():
    rng = np.random.default_rng(2)
    for trial in range(400):
        S = int(rng.choice([2, 4, 8, 16, 33, 64]))
        y = rng.poisson(rng.choice([0.3, 2.0, 8.0, 30.0]), size=S)
        r = float(rng.choice([1e-3, 0.05, 0.9, 1.0, 3.7, 9.5, 9.999, 10.0, 57.0, 1e8]) * rng.uniform(0.9, 1.1))
        nr = int(np.ceil(10.0 - r)) if r < 10.0 else 0
        per_sample = sum(sum(1.0 / (r + i) for i in range(min(int(yj), nr))) for yj in y)
        got = harmonic_row(profile_words(y), r, nr)
        assert abs(got - per_sample) <= 1e-13 * max(1.0, abs(per_sample)), (trial, S, r, nr)
        # ... and the split of the digamma difference it belongs to: H_n + [psi(y + r) - psi(r + n)] = psi(y + r) - psi(r)
        for yj in y[:4]:
            n = min(int(yj), nr)
            lhs = sum(1.0 / (r + i) for i in range(n)) + (special.digamma(yj + r) - special.digamma(r + n))
            rhs = special.digamma(yj + r) - special.digamma(r)
            assert abs(lhs - rhs) <= 1e-9 * max(1.0, abs(rhs)) + 1e-9, (trial, yj, r)


def test_no_contribution_beyond_nr_and_none_when_r_is_large():
    y = np.array([50] * 8)
    w = profile_words(y)
    assert harmonic_row(w, 12.0, 0) == 0.0
    assert harmonic_row(w, 9.5, 1) == 8.0 / 9.5
    full = harmonic_row(w, 0.25, 10)
    assert abs(full - 8 * sum(1.0 / (0.25 + i) for i in range(10))) < 1e-12
