#!/usr/bin/env python3
"""Generate tests/golden/model_golden.npz: outputs of the REFERENCE's CPU model layer.

Pins imcoalhmm_amd/models.py.  The reference's model modules are
Python 2; as in make_fixtures.py a scratch copy under /tmp is converted with the stock ``lib2to3``
tool and imported from there (never committed, never shipped).  Only modules that do not need the
absent ``ziphmm`` are imported.  Everything written is data: parameters in, arrays out.

Run once in the build container:  python tests/golden/make_model_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_fixtures import import_reference_models  # noqa: E402


def hmm(out, key, model, theta):
    pi, T, E = model.build_hidden_markov_model(np.array(theta, dtype=np.float64))
    out[key + "_theta"] = np.array(theta, dtype=np.float64)
    out[key + "_pi"] = np.ascontiguousarray(np.asarray(pi, dtype=np.float64)).reshape(-1)
    out[key + "_T"] = np.ascontiguousarray(np.asarray(T, dtype=np.float64))
    out[key + "_E"] = np.ascontiguousarray(np.asarray(E, dtype=np.float64))


def main():
    IsolationModel, IsolationMigrationModel = import_reference_models()
    from IMCoalHMM import break_points, emissions, state_spaces
    from IMCoalHMM.variable_coalescence_rate_isolation_model import VariableCoalescenceRateIsolationModel
    from IMCoalHMM.variable_migration_model import VariableCoalAndMigrationRateModel
    from IMCoalHMM.isolation_with_migration_model_epochs import IsolationMigrationEpochsModel

    out = {}
    # break points and emission points
    out["bp_exp_7_1.5_0.25"] = np.asarray(break_points.exp_break_points(7, 1.5, 0.25), dtype=np.float64)
    out["bp_exp_20_1000_0.001"] = np.asarray(break_points.exp_break_points(20, 1000.0, 0.001), dtype=np.float64)
    # break_points.trunc_exp_break_points raises TypeError for every input (list + float at
    # break_points.py:58), so the reference holds no values for it
    out["bp_uniform_9_0.5_2.75"] = np.asarray(break_points.uniform_break_points(9, 0.5, 2.75), dtype=np.float64)
    out["bp_psmc_12"] = np.asarray(break_points.psmc_break_points(12), dtype=np.float64)
    out["bp_psmc_8_off"] = np.asarray(break_points.psmc_break_points(8, t_max=7, mu=2e-3, offset=0.125),
                                      dtype=np.float64)
    bp = break_points.exp_break_points(6, 3.0, 0.5)
    out["cp_scalar"] = np.asarray(emissions.coalescence_points(bp, 3.0), dtype=np.float64)
    out["cp_list"] = np.asarray(emissions.coalescence_points(bp, [3.0, 2.0, 1.0, 4.0, 0.5, 2.5]), dtype=np.float64)
    out["emission_cp_scalar"] = np.asarray(emissions.emission_matrix(out["cp_scalar"]), dtype=np.float64)

    # state-space sizes and class counts (B, L, R, E)
    for name, cls in (("isolation", state_spaces.Isolation), ("single", state_spaces.Single),
                      ("migration", state_spaces.Migration)):
        sp = cls()
        out["space_" + name] = np.array([len(sp.states), len(sp.begin_states), len(sp.left_states),
                                         len(sp.right_states), len(sp.end_states), len(sp.transitions)])

    # small and odd-sized HMMs
    for n in (2, 3, 5):
        hmm(out, "iso%d" % n, IsolationModel(n), (0.7, 1.3, 0.4))
    hmm(out, "iso7_fast", IsolationModel(7), (1e-4, 2500.0, 2.5))
    for a, b in ((2, 2), (3, 4), (5, 2)):
        hmm(out, "im_%d_%d" % (a, b), IsolationMigrationModel(a, b), (0.5, 1.0, 1.0, 0.4, 0.1))
    hmm(out, "im_4_6_scaled", IsolationMigrationModel(4, 6), (0.0008, 0.0013, 900.0, 0.7, 350.0))

    hmm(out, "vcr_nosplit", VariableCoalescenceRateIsolationModel([2, 3, 1]), (900.0, 1500.0, 700.0, 0.4))
    hmm(out, "vcr_split", VariableCoalescenceRateIsolationModel([1, 2, 2], est_split=True),
        (2e-9, 1000.0, 400.0, 2200.0, 0.8))
    for cfg, name in ((VariableCoalAndMigrationRateModel.INITIAL_11, "11"),
                      (VariableCoalAndMigrationRateModel.INITIAL_12, "12"),
                      (VariableCoalAndMigrationRateModel.INITIAL_22, "22")):
        hmm(out, "vmig_" + name, VariableCoalAndMigrationRateModel(cfg, [2, 3]),
            (1000.0, 1400.0, 800.0, 1100.0, 150.0, 250.0, 90.0, 300.0, 0.4))
    theta = [0.5, 1.0, 0.4] + [1.0, 1.5, 0.8, 1.2, 0.9] + [0.1, 0.3]
    hmm(out, "epochs_2_3_2", IsolationMigrationEpochsModel(2, 3, 2), theta)

    np.savez_compressed(os.path.join(HERE, "model_golden.npz"), **out)
    for k in sorted(out):
        if not k.endswith(("_T", "_E", "_pi")):
            print(k, out[k] if out[k].size < 8 else out[k].shape)


if __name__ == "__main__":
    main()
