"""Host-side model layer (imcoalhmm_amd/models.py) against outputs of the reference's own classes.

Goldens: tests/golden/hmm_params.npz (make_fixtures.py) and tests/golden/model_golden.npz
(make_model_fixtures.py), both produced by importing the reference's CPU model modules.  The path
is floating point: tolerance 1e-12 absolute on probabilities and 1e-11 relative on T entries
(observed agreement is ~1e-15 / 4e-14; the association order of the matrix products differs).
Values from the reference's own unit tests (tests/IMCoalHMM/break_points_tests.py:39-48,93-98)
are checked bit for bit where they are exact there and to 1e-15 otherwise.
"""
import os

import numpy as np
import pytest

from imcoalhmm_amd import models as M

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "model_golden.npz"))
PARAMS = np.load(os.path.join(HERE, "golden", "hmm_params.npz"))

ABS_TOL = 1e-12
REL_TOL = 1e-11


def check_hmm(gold, key, model):
    pi, T, E = model.build_hidden_markov_model(gold[key + "_theta"])
    want_pi, want_T, want_E = gold[key + "_pi"], gold[key + "_T"], gold[key + "_E"]
    assert pi.shape == want_pi.shape and T.shape == want_T.shape and E.shape == want_E.shape
    assert pi.dtype == np.float64 and T.dtype == np.float64 and E.dtype == np.float64
    assert np.abs(pi - want_pi).max() < ABS_TOL
    assert np.abs(T - want_T).max() < ABS_TOL
    nz = want_T > 1e-200
    assert np.abs(T[nz] / want_T[nz] - 1).max() < REL_TOL
    assert np.abs(E - want_E).max() < ABS_TOL
    assert abs(pi.sum() - 1) < 1e-12 and np.abs(T.sum(axis=1) - 1).max() < 1e-12


def test_break_points_reference_unit_test_values():
    # tests/IMCoalHMM/break_points_tests.py:39-48
    want5 = [0.0, 0.22314355131420976, 0.51082562376599072, 0.916290731874155, 1.6094379124341005]
    assert list(M.exp_break_points(5, 1.0)) == want5
    want10 = [-100.0, -99.947319742171089, -99.888428224342888, -99.821662528030629, -99.744587188117009,
              -99.653426409720026, -99.541854634062929, -99.398013597837036, -99.195281043782956,
              -98.848707453502982]
    assert list(M.exp_break_points(10, 2.0, -100.0)) == want10
    for n in range(1, 50):
        assert len(M.exp_break_points(n, 1.0)) == n
        assert len(M.uniform_break_points(n, 1.0, 2.0)) == n
        assert len(M.psmc_break_points(n)) == n
    # break_points_tests.py:93-98 style: start included, end excluded, equidistant
    pts = M.uniform_break_points(4, 1.0, 3.0)
    assert list(pts) == [1.0, 1.5, 2.0, 2.5]


@pytest.mark.parametrize("key,call", [
    ("bp_exp_7_1.5_0.25", lambda: M.exp_break_points(7, 1.5, 0.25)),
    ("bp_exp_20_1000_0.001", lambda: M.exp_break_points(20, 1000.0, 0.001)),
    ("bp_uniform_9_0.5_2.75", lambda: M.uniform_break_points(9, 0.5, 2.75)),
    ("bp_psmc_12", lambda: M.psmc_break_points(12)),
    ("bp_psmc_8_off", lambda: M.psmc_break_points(8, t_max=7, mu=2e-3, offset=0.125)),
])
def test_break_points_match_reference(key, call):
    got = np.asarray(call(), dtype=np.float64)
    assert got.shape == GOLD[key].shape
    assert np.abs(got - GOLD[key]).max() <= 1e-15 * max(1.0, np.abs(GOLD[key]).max())


def test_trunc_exp_break_points_definition():
    # the reference's function raises TypeError for every input (break_points.py:58); check the
    # documented meaning instead: equal probability mass of Exp(rate) truncated at `end`
    rate, end = 2.0, 1.5
    pts = M.trunc_exp_break_points(6, rate, end)
    mass = (1 - np.exp(-rate * pts)) / (1 - np.exp(-rate * end))
    assert np.allclose(mass, np.arange(6) / 6.0, atol=1e-14)
    assert np.allclose(M.trunc_exp_break_points(6, rate, end, 0.5), pts + 0.5)


def test_coalescence_points_and_emissions():
    bp = M.exp_break_points(6, 3.0, 0.5)
    assert np.abs(M.coalescence_points(bp, 3.0) - GOLD["cp_scalar"]).max() < 1e-14
    assert np.abs(M.coalescence_points(bp, [3.0, 2.0, 1.0, 4.0, 0.5, 2.5]) - GOLD["cp_list"]).max() < 1e-14
    assert np.abs(M.emission_matrix(GOLD["cp_scalar"]) - GOLD["emission_cp_scalar"]).max() < 1e-15
    with pytest.raises(ValueError):
        M.coalescence_points(bp, [1.0, 2.0])


@pytest.mark.parametrize("name,space", [("isolation", M.isolation_space), ("single", M.single_space),
                                        ("migration", M.migration_space)])
def test_state_space_sizes(name, space):
    sp = space()
    got = [sp.size, len(sp.begin_states), len(sp.left_states), len(sp.right_states), len(sp.end_states),
           len(sp.edges)]
    assert got == list(GOLD["space_" + name])
    Q = sp.rate_matrix({label: 1.0 + k for k, label in enumerate(sp.labels)})
    assert np.abs(Q.sum(axis=1)).max() < 1e-12
    assert (Q - np.diag(np.diag(Q)) >= 0).all()


@pytest.mark.parametrize("n,key", [(2, "iso2"), (3, "iso3"), (5, "iso5"), (7, "iso7_fast")])
def test_isolation_model_small(n, key):
    check_hmm(GOLD, key, M.IsolationModel(n))


@pytest.mark.parametrize("key", ["iso10_t0", "iso10_t1", "iso10_t2", "iso10_t3",
                                 "iso20_t0", "iso20_t1", "iso20_t2", "iso20_t3"])
def test_isolation_model_baseline_sizes(key):
    check_hmm(PARAMS, key, M.IsolationModel(int(key[3:5])))


@pytest.mark.parametrize("a,b,key", [(2, 2, "im_2_2"), (3, 4, "im_3_4"), (5, 2, "im_5_2"), (4, 6, "im_4_6_scaled")])
def test_isolation_migration_model_small(a, b, key):
    check_hmm(GOLD, key, M.IsolationMigrationModel(a, b))


@pytest.mark.parametrize("a,b,key", [(10, 10, "im20_t0"), (10, 10, "im20_t1"), (75, 75, "im150_t0")])
def test_isolation_migration_model_baseline_sizes(a, b, key):
    check_hmm(PARAMS, key, M.IsolationMigrationModel(a, b))


def test_variable_coalescence_rate_model():
    check_hmm(GOLD, "vcr_nosplit", M.VariableCoalescenceRateIsolationModel([2, 3, 1]))
    check_hmm(GOLD, "vcr_split", M.VariableCoalescenceRateIsolationModel([1, 2, 2], est_split=True))


@pytest.mark.parametrize("cfg,key", [(0, "vmig_11"), (1, "vmig_12"), (2, "vmig_22")])
def test_variable_migration_model(cfg, key):
    check_hmm(GOLD, key, M.VariableCoalAndMigrationRateModel(cfg, [2, 3]))
    # the documented rate meaning gives a different (also valid) HMM
    other = M.VariableCoalAndMigrationRateModel(cfg, [2, 3], reference_rate_order=False)
    pi, T, _ = other.build_hidden_markov_model(GOLD[key + "_theta"])
    assert abs(pi.sum() - 1) < 1e-12 and np.abs(T - GOLD[key + "_T"]).max() > 1e-6
    with pytest.raises(ValueError):
        M.VariableCoalAndMigrationRateModel(7, [2, 3])


def test_epochs_model():
    check_hmm(GOLD, "epochs_2_3_2", M.IsolationMigrationEpochsModel(2, 3, 2))
    with pytest.raises(ValueError):
        M.IsolationMigrationEpochsModel(2, 3, 2).build_hidden_markov_model(np.ones(9))


def test_valid_parameters_and_batch():
    m = M.IsolationModel(6)
    assert m.valid_parameters(np.array([0.001, 1000.0, 0.4]))
    assert not m.valid_parameters(np.array([0.001, -1.0, 0.4]))
    assert not m.valid_parameters(np.array([0.0, 1.0, 0.4]))
    thetas = np.array([[0.001, 1000.0, 0.4], [0.002, 800.0, 0.6], [0.0005, 1500.0, 0.2]])
    pis, Ts, Es = m.build_batch(thetas)
    assert pis.shape == (3, 6) and Ts.shape == (3, 6, 6) and Es.shape == (3, 6, 3)
    for k in range(3):
        pi, T, E = m.build_hidden_markov_model(thetas[k])
        # stacked matrix products may round differently from the B=1 call: agreement to a few ulp
        assert np.abs(pi - pis[k]).max() < 1e-15 and np.abs(T - Ts[k]).max() < 1e-14 and (E == Es[k]).all()


def test_joint_matrix_is_symmetric_in_time_reversible_setting():
    # pi T is the joint matrix J, symmetric by construction (transitions.py:235)
    pi, T, _ = M.IsolationMigrationModel(4, 5).build_hidden_markov_model(np.array([0.3, 0.8, 1.2, 0.5, 0.2]))
    J = pi[:, None] * T
    assert np.abs(J - J.T).max() < 1e-15


def test_build_speed_n150():
    import time
    m = M.IsolationMigrationModel(75, 75)
    theta = np.array([0.001, 0.001, 1000.0, 0.4, 200.0])
    m.build_hidden_markov_model(theta)
    t0 = time.perf_counter()
    for _ in range(5):
        m.build_hidden_markov_model(theta * (1 + 1e-3))
    per = (time.perf_counter() - t0) / 5
    assert per < 0.25, "N=150 HMM build took %.3f s (reference: ~1.6 s)" % per


# ---------------------------------------------------------------------------------------------
# native path (csrc/model_host.hpp, include/imcoal_model.h) against the numpy path
# ---------------------------------------------------------------------------------------------

class _numpy_only(object):
    """Switch the native path off inside a with-block (the golden tests above run through it whenever the library is built)."""
    def __enter__(self):
        self.saved = dict(M._native)
        M._native.update(lib=None, tried=True)

    def __exit__(self, *exc):
        M._native.update(self.saved)


def _native_or_skip():
    if M._native_lib() is None:
        pytest.skip("libimcoal_fwd.so is not built: the numpy path is the only one")


@pytest.mark.parametrize("make,theta", [
    (lambda: M.IsolationModel(10), [0.001, 1000.0, 0.4]),
    (lambda: M.IsolationModel(20), [0.002, 800.0, 0.6]),
    (lambda: M.IsolationModel(2), [0.001, 1000.0, 0.4]),
    (lambda: M.VariableCoalescenceRateIsolationModel([2, 3, 1]), None),
    (lambda: M.VariableCoalescenceRateIsolationModel([1, 2, 2], est_split=True), None),
])
def test_native_and_numpy_paths_agree(make, theta):
    _native_or_skip()
    model = make()
    if theta is None:
        key = "vcr_split" if getattr(model, "est_split", False) else "vcr_nosplit"
        theta = GOLD[key + "_theta"]
    theta = np.asarray(theta, dtype=np.float64)
    thetas = np.stack([theta * (1.0 + 0.03 * k) for k in range(9)])
    pi_n, T_n, E_n = model.build_hidden_markov_model(theta)
    pis_n, Ts_n, Es_n = model.build_batch(thetas)
    assert M._native["structures"], "the native path did not run"
    with _numpy_only():
        pi_p, T_p, E_p = model.build_hidden_markov_model(theta)
        pis_p, Ts_p, Es_p = model.build_batch(thetas)
    assert np.abs(pi_n - pi_p).max() < 1e-13 and np.abs(T_n - T_p).max() < 1e-13 and (E_n == E_p).all()
    assert np.abs(pis_n - pis_p).max() < 1e-13 and np.abs(Ts_n - Ts_p).max() < 1e-13 and (Es_n == Es_p).all()
    assert np.abs(pis_n[0] - pi_n).max() < 1e-15 and np.abs(Ts_n[0] - T_n).max() < 1e-15     # a system's result does not depend on its batch


def test_native_expm_matches_scipy():
    _native_or_skip()
    import scipy.linalg
    rng = np.random.default_rng(5)
    for n in (1, 2, 4, 15, 31):
        for scale in (1e-4, 1e-2, 0.2, 0.9, 2.0, 2.2, 4.5, 5.3, 6.0, 40.0, 900.0):     # every Pade degree and the scaled branch
            Q = rng.random((n, n)) * scale / n
            np.fill_diagonal(Q, 0.0)
            np.fill_diagonal(Q, -Q.sum(axis=1))
            got = M.expm(Q)
            assert np.abs(got - scipy.linalg.expm(Q)).max() < 5e-13, (n, scale)
            assert np.abs(got.sum(axis=1) - 1).max() < 1e-12


def test_large_state_spaces_stay_on_numpy():
    _native_or_skip()
    before = len(M._native["structures"])
    m = M.IsolationMigrationModel(3, 3)                       # 94-state migration space > NATIVE_MAX_SPACE
    m.build_hidden_markov_model(np.array([0.001, 0.001, 1000.0, 0.4, 200.0]))
    new = list(M._native["structures"].values())[before:]
    assert all(st is False for st in new)


def test_start_outside_the_begin_class_is_refused_on_both_paths():
    sp = M.single_space()
    Q = sp.rate_matrix(M.single_rates(1000.0, 0.4))
    bad = np.zeros(sp.size)
    bad[sp.end_states[0]] = 1.0
    system = M.PiecewiseCTMC(bad, [sp] * 3, [(Q, 1e-3, None), (Q, 2e-3, None)])
    with pytest.raises(ValueError):
        M.hmm_transitions(system)
    with _numpy_only():
        with pytest.raises(ValueError):
            M.hmm_transitions(system)
