"""Full-size (BASELINE config 2: 20 states, 1 x 1e8 columns) checks through size-independent
properties, since the CPU oracle needs minutes at this size:

* closed form: with rank-one T the log-likelihood is a function of the symbol counts only;
* segmentation invariance: two different parallel-in-time splits agree to ~1e-13;
* oracle spot check: the first 2e6 columns as their own chunk against the CPU oracle.
"""
import math

import numpy as np
import pytest

from conftest import rel_err
from imcoalhmm_amd import Forwarder, _capi, synth

pytestmark = pytest.mark.gpu
L_FULL = 100_000_000


@pytest.fixture(scope="module")
def big(hmm_params):
    pi, T, E = hmm_params("iso20_t0")
    parts = [synth.sample_alignment(pi, T, E, 10_000_000, seed=20240001 + k) for k in range(10)]
    obs = np.concatenate(parts)
    assert obs.size == L_FULL
    return obs, Forwarder.from_array(obs, 3)


def test_full_size_closed_form(big):
    obs, f = big
    n, nsym = 20, 3
    rng = np.random.default_rng(11)
    q = rng.random(n); q /= q.sum()
    T = np.tile(q, (n, 1))
    E = rng.random((n, nsym)); E /= E.sum(axis=1, keepdims=True)
    pi = rng.random(n); pi /= pi.sum()
    cnt = np.bincount(obs[1:], minlength=nsym)
    want = math.log(pi @ E[:, obs[0]]) + float(cnt @ np.log(q @ E))
    got = f.forward(pi, T, E)
    assert rel_err(got, want) < 1e-10, (got, want)


def test_full_size_segmentation_invariance_and_spot_check(big, hmm_params, oracle):
    obs, f = big
    pi, T, E = hmm_params("iso20_t0")
    L = _capi.lib()
    try:
        L.imc_set_segment_length(0)
        a = f.forward(pi, T, E)
        L.imc_set_segment_length(50_000)
        b = f.forward(pi, T, E)
    finally:
        L.imc_set_segment_length(0)
    assert math.isfinite(a) and rel_err(a, b) < 1e-12, (a, b)
    # the per-column kernel on the same 1e8 columns (compression off at creation) agrees too
    try:
        L.imc_set_compression(0)
        c = Forwarder.from_array(obs, 3).forward(pi, T, E)
    finally:
        L.imc_set_compression(1)
    assert rel_err(a, c) < 1e-12, (a, c)
    head = obs[:2_000_000]
    got = Forwarder.from_array(head, 3).forward(pi, T, E)
    assert rel_err(got, oracle.forward_scaled(pi, T, E, head)) < 1e-11
