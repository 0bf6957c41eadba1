"""tools/replay_ranks.py ReadTable: the quota rule (pbsim.cpp:3792-3800) restated on a table of per-read (raw length, pass-0
bases) with prefix sums and a binary search -- against the rule walked read by read, on random tables.  (The GPU tests check
the virtual ranks against job.cpp at every exchange; this one checks the restatement against the reference's loop.)"""
import os
import sys

import numpy as np
import pytest

import harness

sys.path.insert(0, os.path.join(harness.ROOT, "tools"))


def sequential(rawlen, out0, first, n, before, quota):
    """the reference's loop: a read is drawn while len_total < quota; the first one whose raw length would pass the quota is
    cut to what is left -- there the block's full-length reads end (the truncated tail reads are walked apart, one after the
    other); otherwise its pass-0 bases are added.  -> (full-length reads, a truncated read is due, len_total)"""
    t, k = before, 0
    while k < n:
        if t >= quota:
            break
        if t + rawlen[first - 1 + k] > quota:          # pbsim.cpp:3795-3800
            break
        t += out0[first - 1 + k]
        k += 1
    return k, int(k < n and t < quota), t


@pytest.mark.parametrize("seed", range(6))
def test_cut_equals_the_loop(seed):
    import replay_ranks as RR
    rng = np.random.default_rng(seed)
    m = 5000
    raw = rng.gamma(2.0, 4000.0, m).astype(np.int64) + 100
    out = (raw * rng.uniform(0.85, 1.05, m)).astype(np.int64)
    tab = RR.ReadTable(raw, out)
    total = int(out.sum())
    for _ in range(300):
        first = int(rng.integers(1, m - 10))
        n = int(rng.integers(1, min(900, m - first + 1)))
        before = int(rng.integers(0, total // 2))
        span = tab.block_sum(first, n)
        quota = before + int(rng.integers(-1000, span + 20000))
        assert tab.cut(first, n, before, quota) == sequential(raw, out, first, n, before, quota), (first, n, before, quota)


def test_block_sum_refuses_reads_beyond_the_table():
    import replay_ranks as RR
    tab = RR.ReadTable(np.array([10, 20, 30]), np.array([9, 19, 29]))
    assert tab.block_sum(2, 2) == 48
    with pytest.raises(RuntimeError):
        tab.block_sum(2, 3)
