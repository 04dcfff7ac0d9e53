"""Batched particle swarm (imcoalhmm_amd/pso.py): update rule, exit conditions and the one-call-per-iteration contract."""
import datetime

import numpy as np
import pytest

from imcoalhmm_amd import pso


def test_swarm_finds_the_maximum_with_one_batch_call_per_iteration():
    target = np.array([0.3, 0.7, 0.55])
    calls = []

    def fitness(x):
        calls.append(x.shape)
        return -np.sum((x - target) ** 2, axis=1)

    opt = pso.Optimiser(seed=4)
    opt.particle_count = 40
    opt.max_iterations = 120
    ctx = opt.maximise(fitness, 3)
    assert ctx.exit_condition == pso.ExitCondition.ITERATIONS and ctx.iteration == 120
    assert len(calls) == 120 and all(c == (40, 3) for c in calls)        # initial batch + 119 iterations
    assert ctx.evaluations == 120 * 40
    assert np.abs(ctx.swarm_best_position - target).max() < 1e-3
    assert ctx.swarm_best_fitness == ctx.best_fitness.max() and ctx.swarm_best_fitness > -1e-5


def test_nan_fitness_is_minus_infinity_and_never_best():
    def fitness(x):
        v = -np.abs(x[:, 0] - 0.5)
        v[x[:, 0] > 0.6] = np.nan                                        # particle_swarm.py:113-117
        return v

    opt = pso.Optimiser(seed=1)
    opt.particle_count = 30
    opt.max_iterations = 30
    ctx = opt.maximise(fitness, 1)
    assert np.isfinite(ctx.swarm_best_fitness) and ctx.swarm_best_position[0] <= 0.6
    assert not np.isnan(ctx.fitness).any()


def test_abort_and_timeout_and_errors():
    opt = pso.Optimiser(seed=2)
    opt.particle_count = 5
    seen = []

    def log(ctx):
        seen.append(ctx.iteration)
        if ctx.iteration == 3:
            ctx.aborted = True

    opt.log = log
    ctx = opt.maximise(lambda x: -x[:, 0] ** 2, 2)
    assert ctx.exit_condition == pso.ExitCondition.ABORT and seen == [1, 2, 3]
    opt2 = pso.Optimiser(seed=3)
    opt2.particle_count = 5
    opt2.max_iterations = None
    opt2.timeout = datetime.timedelta(seconds=0)
    assert opt2.maximise(lambda x: -x[:, 0] ** 2, 1).exit_condition == pso.ExitCondition.TIMEOUT
    with pytest.raises(ValueError):
        opt2.maximise(lambda x: np.zeros(3), 1)                          # wrong number of fitness values
    with pytest.raises(ValueError):
        opt2.maximise(None, 1)


def test_velocity_update_rule():
    """One iteration by hand: v' = omega v + phi_p r_p (best_p - x) + phi_s r_s (best_s - x), x' = x + v'."""
    opt = pso.Optimiser(seed=9)
    opt.particle_count = 4
    opt.max_iterations = 2
    snapshots = []
    opt.log = lambda ctx: snapshots.append((ctx.positions.copy(), ctx.velocities.copy(), ctx.best_positions.copy(),
                                            ctx.swarm_best_position.copy()))
    ctx = opt.maximise(lambda x: -np.sum(x ** 2, axis=1), 2)
    x0, v0, b0, s0 = snapshots[0]
    rng = np.random.default_rng(9)
    rng.uniform(0.0, 1.0, size=(4, 2)); rng.uniform(-0.02, 0.02, size=(4, 2))     # the two initial draws
    rp = rng.uniform(0.0, 1.0, size=(4, 1)); rs = rng.uniform(0.0, 1.0, size=(4, 1))
    v1 = 0.9 * v0 + 0.3 * rp * (b0 - x0) + 0.1 * rs * (s0[None, :] - x0)
    assert np.allclose(ctx.velocities, v1, rtol=0, atol=1e-15) and np.allclose(ctx.positions, x0 + v1, rtol=0, atol=1e-15)
