"""Independent second opinions for the oracle (TEST INFRASTRUCTURE ONLY).

* forward_loglik: numpy textbook scaled forward (semantics of
  src/IMCoalHMM/hmm.py:19-21).  Different summation order from the C oracle (BLAS dot).
* brute_force_loglik: explicit enumeration of all N**L hidden paths (L <= ~10), the known-answer
  arbiter for tiny HMMs.
"""
import itertools
import math

import numpy as np


def forward_loglik(pi, T, E, obs):
    pi = np.asarray(pi, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    E = np.asarray(E, dtype=np.float64)
    obs = np.asarray(obs)
    if obs.size == 0:
        return 0.0
    Tt = np.ascontiguousarray(T.T)
    Et = np.ascontiguousarray(E.T)
    a = pi * Et[obs[0]]
    c = a.sum()
    if not c > 0:
        return -math.inf if c == 0 else math.nan
    a = a / c
    ll = math.log(c)
    for o in obs[1:]:
        a = (Tt @ a) * Et[o]
        c = a.sum()
        if not c > 0:
            return -math.inf if c == 0 else math.nan
        a = a / c
        ll += math.log(c)
    return ll


def brute_force_loglik(pi, T, E, obs):
    pi = np.asarray(pi, dtype=np.float64)
    T = np.asarray(T, dtype=np.float64)
    E = np.asarray(E, dtype=np.float64)
    N, L = len(pi), len(obs)
    if L == 0:
        return 0.0
    total = 0.0
    for path in itertools.product(range(N), repeat=L):
        p = pi[path[0]] * E[path[0], obs[0]]
        for t in range(1, L):
            p *= T[path[t - 1], path[t]] * E[path[t], obs[t]]
        total += p
    return math.log(total) if total > 0 else -math.inf
