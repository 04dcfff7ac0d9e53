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


def test_full_size_150_states_handoff_on_off_and_closed_form(big, hmm_params):
    """BASELINE config 3 (150 states, the same 1e8 columns): the certified rank-one hand-off changes the work, not the
    value - on/off agree to 1e-13 relative, repeated calls are bit-identical (the hand-off point is decided per
    segment from the data, nothing adapts between calls), and a rank-one T reproduces the count-only closed form
    through the same kernels."""
    obs, f = big
    pi, T, E = hmm_params("im150_t0")
    L = _capi.lib()
    try:
        _capi.check(L.imc_set_rank1_handoff(1))
        on = [f.forward(pi, T, E) for _ in range(4)]
        stats = _capi.last_rank1()
        kernels = _capi.last_plan()["kernels"]
        _capi.check(L.imc_set_rank1_handoff(0))
        off = f.forward(pi, T, E)
        assert "rank1" not in _capi.last_plan()["kernels"]
    finally:
        _capi.check(L.imc_set_rank1_handoff(1))
    assert "rank1-handoff" in kernels and stats[0] > 100 and stats[1] > 0, (kernels, stats)
    assert len(set(on)) == 1, on          # no state is carried between calls: repeated evaluations are bit-identical
    assert math.isfinite(off) and all(rel_err(v, off) < 1e-13 for v in on), (on, off)
    n, nsym = 150, 3
    rng = np.random.default_rng(12)
    q = rng.random(n); q /= q.sum()
    Tq = np.tile(q, (n, 1))
    Eq = rng.random((n, nsym)); Eq /= Eq.sum(axis=1, keepdims=True)
    piq = rng.random(n); piq /= piq.sum()
    cnt = np.bincount(obs[1:], minlength=nsym)
    want = math.log(piq @ Eq[:, obs[0]]) + float(cnt @ np.log(q @ Eq))
    got = f.forward(piq, Tq, Eq)
    assert _capi.last_rank1()[1] == _capi.last_rank1()[0] > 0       # a rank-one T collapses at once
    assert rel_err(got, want) < 1e-10, (got, want)
