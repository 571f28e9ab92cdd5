"""CPU model of k_ransac_samples' next(i) table (openpano_amd/csrc/ransac.hip, rs_next_table).

The reference draws sample indices one by one and rejects repeats until it holds ns distinct ones
(transform_estimate.cc:70-77), so hypothesis k + 1 starts where hypothesis k's rejections let it.  The kernel computes, for
EVERY position i of a chunk of the reduced mt19937 stream, next(i) = the position right after the sample that starts at i,
and finds the hypothesis starts by pointer jumping.  next(i) is computed without walking a sample per position: a thread
keeps the distinct values to the right of i in first-occurrence order and moves rd[i] to the front as i steps left; the
position of the list's ns-th entry, plus one, is next(i).  This file restates that procedure step for step (segment walk,
forward start list, the chunk-end case) and checks it against the sequential rejection loop on random streams -- the
GPU tests check the kernel itself against the oracle, this one documents and pins the algorithm where no GPU is needed.
"""
import numpy as np
import pytest

END = 0xFFFF
SEG = 24          # RS_SEG: start positions per thread


def next_sequential(rd, ns):
    """next(i) by the reference's loop: take draws from i on, skip values already taken, stop at ns distinct."""
    n = len(rd)
    out = np.full(n + 1, END, dtype=np.int64)
    for i in range(n):
        seen = set()
        for t in range(i, n):
            if rd[t] not in seen:
                seen.add(rd[t])
                if len(seen) == ns:
                    out[i] = t + 1
                    break
    return out


def next_move_to_front(rd, ns):
    """rs_next_table: per segment [a0, b) of SEG starts, right to left, one move-to-front per start."""
    n = len(rd)
    out = np.full(n + 1, END, dtype=np.int64)
    for a0 in range(0, n, SEG):
        b = min(a0 + SEG, n)
        val, pos, cnt = [-1] * 8, [0] * 8, 0

        def to_front(x, at):
            nonlocal cnt
            pv, pp = val[0], pos[0]
            c, absent = True, True
            val[0], pos[0] = x, at
            for q in range(1, 8):
                c = c and pv != x
                if q == ns:
                    absent = c
                tv, tp = val[q], pos[q]
                if c:
                    val[q], pos[q] = pv, pp
                pv, pp = tv, tp
            if ns == 8:
                absent = c and pv != x
            if absent and cnt < ns:
                cnt += 1

        # the list at b: the sample that starts at b, walked forward, newest first ...
        t = b
        while t < n and cnt < ns:
            x = rd[t]
            if all(v != x for v in val):
                val[1:], pos[1:] = val[:-1], pos[:-1]
                val[0], pos[0] = x, t
                cnt += 1
            t += 1
        if cnt == ns:       # ... then turned round into first-occurrence order
            val[:ns], pos[:ns] = val[:ns][::-1], pos[:ns][::-1]
        else:               # the chunk ended first: the right-to-left walk from the chunk's end
            val, pos, cnt = [-1] * 8, [0] * 8, 0
            for i in range(n - 1, b - 1, -1):
                to_front(int(rd[i]), i)
        for i in range(b - 1, a0 - 1, -1):
            to_front(int(rd[i]), i)
            out[i] = pos[ns - 1] + 1 if cnt >= ns else END
    return out


@pytest.mark.parametrize("ns", [8, 7, 5])
def test_move_to_front_equals_the_rejection_loop(ns):
    rng = np.random.default_rng(1234 + ns)
    cases = [(8, 700), (9, 500), (10, 1500), (13, 900), (30, 600), (64, 400), (65, 400), (213, 800), (4000, 300)]
    for m, n in cases:
        if m < ns:
            continue
        for rep in range(3):
            rd = rng.integers(0, m, size=n + rep * 7)
            a, b = next_sequential(rd, ns), next_move_to_front(rd, ns)
            assert np.array_equal(a, b), (m, n, rep, np.flatnonzero(a != b)[:8])


def test_short_and_degenerate_chunks():
    for rd in ([], [3], [1, 1, 1, 1], list(range(8)), list(range(8)) * 2, [5] * 30 + list(range(8)), list(range(7)) * 9):
        rd = np.array(rd, dtype=np.int64)
        assert np.array_equal(next_sequential(rd, 8), next_move_to_front(rd, 8))


def test_next_is_monotone_and_chains_give_the_reference_samples():
    """The hypothesis starts are next^k(0); walking them reproduces the reference's consecutive samples."""
    rng = np.random.default_rng(7)
    m, ns, n = 11, 8, 4000
    rd = rng.integers(0, m, size=n)
    nx = next_move_to_front(rd, ns)
    valid = nx[:n][nx[:n] != END]
    assert np.all(np.diff(valid) >= 0)
    # sequential automaton
    starts, i = [], 0
    while True:
        seen, t = [], i
        while t < n and len(seen) < ns:
            if rd[t] not in seen:
                seen.append(rd[t])
            t += 1
        if len(seen) < ns:
            break
        starts.append(i); i = t
    chain, i = [], 0
    while i < n and nx[i] != END:
        chain.append(i); i = int(nx[i])
    assert chain == starts
