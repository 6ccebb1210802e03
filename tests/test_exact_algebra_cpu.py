"""The identities behind the prover's shortcuts, checked on the CPU oracle.

TEST INFRASTRUCTURE.  Field arithmetic is exact, so a rearrangement that is an identity of polynomials gives the same proof bits; the
GPU suite checks the bits (options on and off against the oracle's plain evaluation), this file checks the mathematics the options
rest on, with the oracle's own transforms:
  * the first half of the bit-reversed 2N-point extension is an N-point circle domain whose twiddles are the first halves of the 2N
    tables' layers (what fft.hip twiddles_first_half builds);
  * the quotient of a degree-2 constraint is Q0 + t Z: one coefficient beyond the N-point space, at index N;
  * t follows from the first half and ONE further row; a degree-3 quotient fills the 2N-point space, a degree-4 one does not fit it;
  * a linear combination of columns can be taken in coefficient space (the DEEP-quotient numerators)."""
import numpy as np
import pytest

import oracle_lib as O
from test_air_program_cpu import denominators

P = O.P


def _tables(T):
    tw, itw = T.arrays()
    return [int(x) for x in tw], [int(x) for x in itw]


def sub_interpolate(v, n, ntw, itw):
    """iFFT of 2^n bit-reversed values with the tables of a size-2^ntw transform: ntw = n is the canonic coset (oracle/poly.h interpolate),
    ntw = n + 1 the first half of the 2^(n+1)-point domain."""
    TL = len(itw)
    line = lambda layer, h: itw[TL - (1 << (ntw - layer)) + h]

    def circ(h):
        c = h >> 2
        x, y = line(1, 2 * c), line(1, 2 * c + 1)
        return [y, (P - y) % P, (P - x) % P, x][h & 3]
    v = [int(x) for x in v]
    N = 1 << n
    for h in range(N // 2):
        t, a, b = circ(h), v[2 * h], v[2 * h + 1]
        v[2 * h], v[2 * h + 1] = (a + b) % P, (a - b) * t % P
    for layer in range(1, n):
        for h in range(1 << (n - 1 - layer)):
            t = line(layer, h)
            for l in range(1 << layer):
                i0 = (h << (layer + 1)) + l
                i1 = i0 + (1 << layer)
                a, b = v[i0], v[i1]
                v[i0], v[i1] = (a + b) % P, (a - b) * t % P
    inv = pow(N, P - 2, P)
    return [x * inv % P for x in v]


@pytest.mark.parametrize("n", [4, 6, 7])
def test_first_half_of_the_extension_is_an_n_point_domain(oracle, n):
    rng = np.random.default_rng(n)
    N = 1 << n
    T = O.Twiddles(n + 1)
    _, itw = _tables(T)
    co = rng.integers(0, P, N, dtype=np.uint32)
    assert sub_interpolate(T.evaluate(co, n), n, n, itw) == [int(x) for x in co]                  # the python model == the oracle's interpolate
    assert sub_interpolate(T.evaluate(co, n + 1)[:N], n, n + 1, itw) == [int(x) for x in co]      # first half, head-half twiddles


def _vanishing_product_quotient(T, n, factors, rng):
    """C = f_1 ... f_d - c with c the interpolant of the product on the trace domain (so C vanishes there); returns Q = C / Z on the
    2^(n+e) point domain for the smallest e that holds it, as (e, rows)."""
    N = 1 << n
    cols = [rng.integers(0, P, N, dtype=np.uint32) for _ in range(factors)]
    prod = np.ones(N, np.uint64)
    for c in cols:
        prod = (prod * T.evaluate(c, n)) % P
    cc = T.interpolate(prod.astype(np.uint32))
    return cols, cc


@pytest.mark.parametrize("n", [5, 6])
def test_degree_two_quotient_is_q0_plus_t_z(oracle, n):
    rng = np.random.default_rng(10 + n)
    N = 1 << n
    T = O.Twiddles(n + 2)
    _, itw = _tables(O.Twiddles(n + 1))
    (a, b), cc = _vanishing_product_quotient(T, n, 2, rng)
    den = denominators(n, n + 1)
    A, B, Cc = T.evaluate(a, n + 1), T.evaluate(b, n + 1), T.evaluate(cc, n + 1)
    Q = [((int(A[r]) * int(B[r]) - int(Cc[r])) % P) * int(den[r >> n]) % P for r in range(2 * N)]
    full = [int(x) for x in T.interpolate(np.array(Q, np.uint32))]
    assert [i for i in range(N, 2 * N) if full[i]] == [N]                      # exactly ONE coefficient beyond the N-point space: Z's
    t = full[N]
    # from the first half and one further row
    I = sub_interpolate(Q[:N], n, n + 1, itw)
    z0, z1 = pow(int(den[0]), P - 2, P), pow(int(den[1]), P - 2, P)
    w = N
    Iw = int(T.evaluate(np.array(I, np.uint32), n + 1)[w])
    assert (Q[w] - Iw) * pow((z1 - z0) % P, P - 2, P) % P == t
    I[0] = (I[0] - t * z0) % P
    assert I == full[:N]


def test_degree_three_fills_and_degree_four_overflows_the_2n_space(oracle):
    """d <= 2^e + 1: a cubic quotient needs all 2N coefficients (so it cannot go on the half), a quartic one aliases on 2N points."""
    n, rng = 5, np.random.default_rng(3)
    N = 1 << n
    T = O.Twiddles(n + 2)
    for d, fits in ((3, True), (4, False)):
        cols, cc = _vanishing_product_quotient(T, n, d, rng)
        out = []
        for e in (1, 2):
            den = denominators(n, n + e)
            ev = [T.evaluate(c, n + e) for c in cols]
            Cc = T.evaluate(cc, n + e)
            Q = []
            for r in range(N << e):
                p = 1
                for v in ev:
                    p = p * int(v[r]) % P
                Q.append((p - int(Cc[r])) % P * int(den[r >> n]) % P)
            out.append([int(x) for x in T.interpolate(np.array(Q, np.uint32))])
        small, big = out
        assert not any(big[4 * N // 2 * 0 + i] for i in range(3 * N + 1 if d == 4 else 2 * N, 4 * N))   # cubic: 2N coefficients; quartic: 3N + 1
        assert (small == big[:2 * N] and not any(big[2 * N:])) == fits
        if d == 3:
            assert any(big[N + 1:2 * N])                                           # more than Q0 + t Z: the half-domain trick stops at degree 2


def test_linear_combinations_commute_with_extension(oracle):
    """Σ c_k f_k(d) for secure c_k is, coordinate by coordinate, the extension of the combination of the coefficient columns."""
    n, rng = 6, np.random.default_rng(5)
    N = 1 << n
    T = O.Twiddles(n + 1)
    cols = [rng.integers(0, P, N, dtype=np.uint32) for _ in range(7)]
    cks = rng.integers(0, P, (7, 4), dtype=np.uint32)
    ext = [T.evaluate(c, n + 1).astype(object) for c in cols]
    for q in range(4):
        rows = sum(int(cks[k][q]) * ext[k] for k in range(7)) % P
        comb = sum(int(cks[k][q]) * cols[k].astype(object) for k in range(7)) % P
        assert np.array_equal(np.array(rows, np.uint32), T.evaluate(np.array(comb, np.uint32), n + 1))


def test_degree_four_quotient_from_3n_plus_1_samples(oracle):
    """The next lever (not built in the product): f = f0 + Z2 (f10 + t Z) for a degree-4 quotient — f0 from the
    committed 2N rows, f10 from the first QUARTER of the 4N-point domain (itself an N-point domain: the first quarters of the 4N tables'
    layers), t from one further row; 3N + 1 constraint evaluations instead of 4N, and an N-point transform per column instead of a 4N one."""
    n, rng = 5, np.random.default_rng(9)
    N = 1 << n
    T = O.Twiddles(n + 2)
    _, itw = _tables(T)
    cols, cc = _vanishing_product_quotient(T, n, 4, rng)
    inv = lambda x: pow(int(x) % P, P - 2, P)

    def quotient_rows(e, rows):
        den = denominators(n, n + e)
        ev = [T.evaluate(c, n + e) for c in cols]
        Cc = T.evaluate(cc, n + e)
        out = []
        for r in rows:
            p = 1
            for v in ev:
                p = p * int(v[r]) % P
            out.append((p - int(Cc[r])) % P * int(den[r >> n]) % P)
        return out
    full = [int(x) for x in T.interpolate(np.array(quotient_rows(2, range(4 * N)), np.uint32))]      # Stwo's way: all 4N rows
    # the first quarter of the 4N-point domain is an N-point domain (first quarters of the layers)
    assert sub_interpolate(T.evaluate(cols[0], n + 2)[:N], n, n + 2, itw) == [int(x) for x in cols[0]]
    # 1. the committed 2N rows -> f0
    f0 = T.interpolate(np.array(quotient_rows(1, range(2 * N)), np.uint32))
    # 2. N rows of the first quarter + row N
    f0_4n = T.evaluate(f0, n + 2)
    z2 = inv(denominators(n + 1, n + 2)[0])                       # Z2 = vanishing polynomial of the committed domain: constant on the first half
    d4 = denominators(n, n + 2)
    z0, z1 = inv(d4[0]), inv(d4[1])                               # Z on the first and the second quarter
    fq = quotient_rows(2, range(N + 1))
    g = [(fq[r] - int(f0_4n[r])) % P * inv(z2) % P for r in range(N + 1)]
    I = sub_interpolate(g[:N], n, n + 2, itw)
    Iw = int(T.evaluate(np.array(I, np.uint32), n + 2)[N])
    t = (g[N] - Iw) * inv(z1 - z0) % P
    I[0] = (I[0] - t * z0) % P
    assert [int(x) for x in f0] + I + [t] + [0] * (N - 1) == full


def test_neighbour_rows_of_the_first_quarter_are_the_second_quarter(oracle):
    """The next lever of the quarter-domain composition: a constraint of degree 4 / 5 that reads a neighbour row
    (mask offset +-1) is kept on the 4N-point domain today.  The first quarter of the bit-reversed 4N-point domain is +-(g + <g_{N/2}>)
    with g of order 8N; the trace step has order N, and g + g_N lands in the coset whose bit-reversed positions are the SECOND quarter
    (index bits 10 below the top), an even multiple of the step back in the first.  So the neighbour reads of the quarter-domain rows
    need one more N-point transform per neighbour-read column (the second quarter), not the 4N-point extension.  Checked with the
    oracle's own evaluator: the values a LOAD at offset +-1 / +-2 sees on rows [0, N) of the 4N-point domain are exactly the column's
    extension values at rows [N, 2N) / [0, N), each once."""
    import nexus_zkvm_amd.air_program as ap
    for n in (4, 6):
        N = 1 << n
        rng = np.random.default_rng(n)
        T = O.Twiddles(n + 2)
        co = T.interpolate(rng.integers(0, P, N, dtype=np.uint32))
        E = T.evaluate(co, n + 2)
        assert len(set(int(v) for v in E)) == 4 * N          # distinct values: positions can be read off
        pos = {int(v): i for i, v in enumerate(E)}
        for off, quarter in ((1, 1), (-1, 1), (2, 0), (-2, 0), (3, 1)):
            b = ap.ProgramBuilder()
            (x,) = b.next_trace_mask(0, (off,))
            b.add_constraint(x)
            seen = O.eval_constraint_program(b.build(), [E], [1, 0, 0, 0], np.ones(4 * N, np.uint32), n, n + 2)[0]
            idx = [pos[int(v)] for v in seen[:N]]
            assert sorted(idx) == list(range(quarter * N, (quarter + 1) * N)), (n, off)
