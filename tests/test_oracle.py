"""CPU tests of the oracle itself: it must be pinned before it is trusted as the checker.

The reference holds no golden vector for the forward path, so the pin is:
brute-force path enumeration on tiny HMMs, four independent implementations agreeing on the
reference's example alignment, and the frozen values in tests/golden/loglik_golden.json.
"""
import math

import numpy as np
import pytest

from conftest import rel_err
from oracle import forward_numpy
from imcoalhmm_amd import synth


@pytest.mark.parametrize("n,nsym,L,seed", [(2, 2, 6, 1), (2, 3, 8, 2), (3, 3, 7, 3), (3, 2, 8, 4), (4, 3, 5, 5)])
def test_oracle_matches_brute_force(oracle, n, nsym, L, seed):
    pi, T, E = synth.random_hmm(n, nsym, seed, stay=0.6)
    rng = np.random.default_rng(seed)
    for _ in range(5):
        obs = rng.integers(0, nsym, size=L).astype(np.uint8)
        want = forward_numpy.brute_force_loglik(pi, T, E, obs)
        assert rel_err(oracle.forward_scaled(pi, T, E, obs), want) < 1e-13
        assert rel_err(oracle.forward_scaled_ld(pi, T, E, obs), want) < 1e-13
        assert rel_err(forward_numpy.forward_loglik(pi, T, E, obs), want) < 1e-13
        assert rel_err(oracle.zip_forward_from_raw(pi, T, E, obs, min_count=2), want) < 1e-13


def test_hand_computed_two_state():
    # N=2, L=2, by hand: sum over 4 paths
    pi = np.array([0.6, 0.4]); T = np.array([[0.7, 0.3], [0.2, 0.8]]); E = np.array([[0.9, 0.1], [0.5, 0.5]])
    obs = np.array([0, 1], dtype=np.uint8)
    p = (0.6 * 0.9 * 0.7 * 0.1 + 0.6 * 0.9 * 0.3 * 0.5 + 0.4 * 0.5 * 0.2 * 0.1 + 0.4 * 0.5 * 0.8 * 0.5)
    assert abs(forward_numpy.brute_force_loglik(pi, T, E, obs) - math.log(p)) < 1e-15
    from oracle import oracle_lib
    assert abs(oracle_lib.forward_scaled(pi, T, E, obs) - math.log(p)) < 1e-15


def test_golden_values_reproduce(oracle, hmm_params, example_pairs, golden_loglik):
    """Frozen example-data log-likelihoods (config 1 of BASELINE.json and five more pairs)."""
    for key, rec in golden_loglik.items():
        pname, mkey = key.split("|")
        if mkey == "im150_t0" and pname != "hg18__pantro2":
            continue   # N=150 is slow on CPU; one pair is enough here
        pi, T, E = hmm_params(mkey)
        got = oracle.forward_scaled(pi, T, E, example_pairs[pname])
        assert rel_err(got, rec["loglik"]) < 1e-12, key


def test_survey_indicative_values(golden_loglik):
    # SURVEY.md section 8c scratch values (numpy forward in the survey session)
    assert abs(golden_loglik["hg18__pantro2|iso10_t0"]["loglik"] - (-3638.641281132608)) < 1e-8
    assert abs(golden_loglik["hg18__pantro2|iso20_t0"]["loglik"] - (-3631.823689765038)) < 1e-8


def test_example_pair_symbol_counts(example_pairs):
    # hg18/pantro2 -> L=65,255, counts 62,137 / 642 / 2,476
    o = example_pairs["hg18__pantro2"]
    assert o.size == 65255
    assert np.bincount(o, minlength=3).tolist() == [62137, 642, 2476]


def test_zip_equals_textbook(oracle, hmm_params):
    pi, T, E = hmm_params("iso20_t0")
    obs = synth.sample_alignment(pi, T, E, 200_000, seed=7)
    a = oracle.forward_scaled(pi, T, E, obs)
    z = oracle.Zip(obs, 3)
    assert z.length < obs.size / 4          # it does compress
    assert rel_err(z.forward(pi, T, E), a) < 1e-12
    assert rel_err(oracle.forward_scaled_ld(pi, T, E, obs), a) < 1e-12


def test_chunk_sum_semantics(oracle, hmm_params):
    """likelihood.py:33: every forwarder restarts from pi; values are summed."""
    pi, T, E = hmm_params("iso10_t0")
    chunks = [synth.sample_alignment(pi, T, E, n, seed=10 + k) for k, n in enumerate((1000, 1, 3777, 0, 20))]
    tot, per = oracle.forward_chunks_mt(pi, T, E, chunks, threads=2)
    want = [oracle.forward_scaled(pi, T, E, c) for c in chunks]
    assert per.tolist() == want
    assert tot == sum(want)
    # and it differs from treating the data as one sequence
    joined = np.concatenate(chunks)
    assert abs(oracle.forward_scaled(pi, T, E, joined) - tot) > 1e-6


def test_impossible_sequence_is_minus_inf(oracle):
    pi = np.array([0.5, 0.5]); T = np.array([[0.9, 0.1], [0.1, 0.9]])
    E = np.array([[1.0, 0.0], [1.0, 0.0]])
    obs = np.array([0, 0, 1, 0], dtype=np.uint8)
    assert oracle.forward_scaled(pi, T, E, obs) == -math.inf
    assert forward_numpy.forward_loglik(pi, T, E, obs) == -math.inf
    assert oracle.forward_scaled(pi, T, E, np.zeros(0, dtype=np.uint8)) == 0.0


def test_closed_form_iid_states(oracle):
    """Rank-one T (rows all equal q): loglik has a closed form from the symbol counts."""
    n, nsym = 5, 3
    rng = np.random.default_rng(3)
    q = rng.random(n); q /= q.sum()
    T = np.tile(q, (n, 1))
    E = rng.random((n, nsym)); E /= E.sum(axis=1, keepdims=True)
    pi = rng.random(n); pi /= pi.sum()
    obs = rng.integers(0, nsym, size=5000).astype(np.uint8)
    per_sym = np.log(q @ E)
    want = math.log(pi @ E[:, obs[0]]) + per_sym[obs[1:]].sum()
    assert rel_err(oracle.forward_scaled(pi, T, E, obs), want) < 1e-12
