"""Logup interaction-trace restatement (oracle/logup.h) pinned by the identities the construction
rests on (no Stwo here to compare with): the fractions really are num/den, merged pairs are sums of fractions, the running
column is a sum over columns, the claimed sum is the sum of all fractions, and after finalize_last the column is a prefix
sum in natural coset order whose last row is zero (sum of (x - mean) over all rows)."""
import numpy as np
import pytest

import oracle_lib as O

P = O.P




def q_mul_py(x, y):
    """QM31 = CM31[u]/(u^2 - 2 - i), CM31 = M31[i]/(i^2 + 1); plain Python ints."""
    def cmul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    def cadd(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
    xa, xb, ya, yb = (x[0], x[1]), (x[2], x[3]), (y[0], y[1]), (y[2], y[3])
    bb = cmul(xb, yb)
    r = ((2 * bb[0] - bb[1]) % P, (2 * bb[1] + bb[0]) % P)       # * (2 + i)
    a = cadd(cmul(xa, ya), r)
    b = cadd(cmul(xa, yb), cmul(xb, ya))
    return (a[0], a[1], b[0], b[1])


def q_add_py(x, y):
    return tuple((int(a) + int(b)) % P for a, b in zip(x, y))


def rows(col4):
    return [tuple(int(col4[q][r]) for q in range(4)) for r in range(len(col4[0]))]


def coset_row_position(c, log):
    n = 1 << log
    d = c // 2 if c % 2 == 0 else n - 1 - (c - 1) // 2
    return int(format(d, "0%db" % log)[::-1], 2) if log else 0


@pytest.mark.parametrize("log", [4, 7])
def test_logup_identities(oracle, log):
    rng = np.random.default_rng(log)
    n = 1 << log
    tuple_cols = rng.integers(0, P, (3, n), dtype=np.uint32)
    alphas = rng.integers(0, P, (3, 4), dtype=np.uint32)
    z = rng.integers(0, P, 4, dtype=np.uint32)
    den = oracle.logup_combine(list(tuple_cols), alphas, z)
    for r in (0, 1, n - 1):
        acc = (0, 0, 0, 0)
        for k in range(3):
            acc = q_add_py(acc, q_mul_py(tuple(int(x) for x in alphas[k]), (int(tuple_cols[k][r]), 0, 0, 0)))
        assert rows(den)[r] == tuple((a - int(b)) % P for a, b in zip(acc, z))
    mult = rng.integers(0, 5, n, dtype=np.uint32)
    # one fraction: col * den == num
    col0 = oracle.logup_finalize_col(den, scale_a=(P - 1, 0, 0, 0), mult_a=mult)
    for r in (0, 3, n - 1):
        assert q_mul_py(rows(col0)[r], rows(den)[r]) == ((P - int(mult[r])) % P, 0, 0, 0)
    # two merged fractions + previous column: (col1 - col0) * denA * denB == numA * denB + numB * denA
    den_b = oracle.logup_combine(list(tuple_cols[:2]), alphas[:2], z)
    col1 = oracle.logup_finalize_col(den, den_b=den_b, mult_b=mult, prev=col0)
    for r in (0, 5, n - 1):
        diff = tuple((a - b) % P for a, b in zip(rows(col1)[r], rows(col0)[r]))
        lhs = q_mul_py(q_mul_py(diff, rows(den)[r]), rows(den_b)[r])
        rhs = q_add_py(rows(den_b)[r], q_mul_py((int(mult[r]), 0, 0, 0), rows(den)[r]))
        assert lhs == rhs
    # finalize_last: claimed sum = sum of the column; result = prefix sums of (x - mean) along natural coset order
    last, claimed = oracle.logup_finalize_last(col1)
    total = (0, 0, 0, 0)
    for v in rows(col1):
        total = q_add_py(total, v)
    assert tuple(int(x) for x in claimed) == total
    inv_n = pow(n, P - 2, P)
    mean = tuple(t * inv_n % P for t in total)
    run = (0, 0, 0, 0)
    lr, c1 = rows(last), rows(col1)
    for c in range(n):
        p = coset_row_position(c, log)
        run = q_add_py(run, tuple((a - b) % P for a, b in zip(c1[p], mean)))
        assert lr[p] == run
    assert lr[coset_row_position(n - 1, log)] == (0, 0, 0, 0)
