"""Synthetic pairwise-alignment generator for the bench / parity harness (host side, numpy only).

Follows SURVEY.md section 8d: sample a hidden path from (pi, T), emit symbol 0/1 from
E[:, :2] renormalised, then overwrite ~4 % of the columns with the missing-data symbol 2 in
geometric runs (mean 25), mimicking the 3.8 % missing of the reference's example alignment
(examples/example_data.fa; symbol rule at scripts/prepare-alignments.py:92-105).
No reference code is involved; seeds are fixed by the caller.
"""
import numpy as np


def sample_alignment(pi, T, E, length, seed, missing_frac=0.04, missing_mean_run=25.0):
    """Return a uint8 array of `length` symbols in {0,1,2}."""
    rng = np.random.default_rng(seed)
    pi = np.asarray(pi, dtype=np.float64).reshape(-1)
    T = np.asarray(T, dtype=np.float64)
    E = np.asarray(E, dtype=np.float64)
    n = len(pi)
    length = int(length)
    out = np.empty(length, dtype=np.uint8)
    if length == 0:
        return out
    stay = np.clip(np.diag(T), 0.0, 1.0 - 1e-12)
    jump = T.copy()
    np.fill_diagonal(jump, 0.0)
    rs = jump.sum(axis=1, keepdims=True)
    jump = np.where(rs > 0, jump / np.where(rs > 0, rs, 1.0), 1.0 / n)
    jump_cdf = np.cumsum(jump, axis=1)
    p1 = E[:, 1] / (E[:, 0] + E[:, 1])
    # hidden path as (state, sojourn) runs
    pos = 0
    state = int(rng.choice(n, p=pi / pi.sum()))
    while pos < length:
        # draw a batch of sojourns
        m = 4096
        states = np.empty(m, dtype=np.int64)
        durs = np.empty(m, dtype=np.int64)
        for k in range(m):
            states[k] = state
            durs[k] = rng.geometric(1.0 - stay[state])
            state = int(np.searchsorted(jump_cdf[state], rng.random(), side="right"))
            state = min(state, n - 1)
        cum = np.cumsum(durs)
        keep = int(np.searchsorted(cum, length - pos, side="left")) + 1
        keep = min(keep, m)
        seg_states = np.repeat(states[:keep], durs[:keep])[: length - pos]
        u = rng.random(seg_states.size)
        out[pos:pos + seg_states.size] = (u < p1[seg_states]).astype(np.uint8)
        pos += seg_states.size
        if keep < m:
            break
    # missing data in geometric runs
    if missing_frac > 0:
        n_runs = int(length * missing_frac / missing_mean_run) + 1
        starts = rng.integers(0, length, size=n_runs)
        lens = rng.geometric(1.0 / missing_mean_run, size=n_runs)
        for s, l in zip(starts, lens):
            out[s:s + l] = 2
    return out


def random_hmm(n, nsym, seed, stay=0.999):
    """A generic well-conditioned HMM for parity tests: sticky random T, random E, random pi."""
    rng = np.random.default_rng(seed)
    T = rng.random((n, n)) + 1e-3
    T = T / T.sum(axis=1, keepdims=True) * (1.0 - stay)
    T[np.arange(n), np.arange(n)] += stay
    T = T / T.sum(axis=1, keepdims=True)
    E = rng.random((n, nsym)) + 0.05
    E = E / E.sum(axis=1, keepdims=True)
    pi = rng.random(n) + 0.1
    pi = pi / pi.sum()
    return pi, T, E
