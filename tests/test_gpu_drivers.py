"""SURVEY.md section 8f rank 2 on the device: the batched drivers (MC3 rounds, mcmc.py:165-174; particle swarm,
particle_swarm.py:168-192; genetic algorithm, genetic_algorithm.py:818-831) run through the HIP path and, with the same
seeds, through the same drivers fed by the CPU oracle.  The random streams are identical, so the whole accept / reject
/ swap / breeding trajectory must be identical unless a likelihood differs by more than an acceptance margin; every
batch of log-likelihoods is compared at 1e-11 on the way."""
import random

import numpy as np
import pytest

from conftest import rel_err
from imcoalhmm_amd import Forwarder, Likelihood, _capi, ga, mcmc, models, pso, synth
from imcoalhmm_amd.likelihood import build_hmms

pytestmark = pytest.mark.gpu
TOL = 1e-11


class OracleLikelihood(object):
    """Same surface as imcoalhmm_amd.Likelihood (gate, __call__, batch), values from oracle/ (the checker)."""

    def __init__(self, oracle, model, chunks):
        self.oracle, self.model, self.chunks = oracle, model, chunks

    def batch(self, thetas):
        thetas = [np.asarray(t, dtype=np.float64) for t in thetas]
        out = np.full(len(thetas), -np.inf)
        valid = [k for k, t in enumerate(thetas) if self.model.valid_parameters(t)]
        if valid:
            pis, Ts, Es = build_hmms(self.model, [thetas[k] for k in valid])
            for j, k in enumerate(valid):
                out[k] = self.oracle.forward_chunks_mt(pis[j], Ts[j], Es[j], self.chunks, threads=8)[0]
        return out

    def __call__(self, theta):
        return float(self.batch([theta])[0])


class Recording(object):
    def __init__(self, inner):
        self.inner, self.batches = inner, []

    def batch(self, thetas):
        v = np.asarray(self.inner.batch(thetas), dtype=np.float64)
        self.batches.append(v.copy())
        return v

    def __call__(self, theta):
        return float(self.batch([theta])[0])


def same_batches(dev, ref):
    assert len(dev) == len(ref)
    for a, b in zip(dev, ref):
        assert a.shape == b.shape
        for x, y in zip(a, b):
            assert (x == y) or (np.isfinite(x) and np.isfinite(y) and rel_err(x, y) < TOL), (x, y)


def make_data(model, theta, n_chunks, cols, seed):
    pi, T, E = model.build_hidden_markov_model(np.asarray(theta, dtype=np.float64))
    return [synth.sample_alignment(pi, T, E, cols, seed=seed + k) for k in range(n_chunks)]


def test_mc3_64_chains_trajectory(oracle):
    """BASELINE config[4]'s evaluation stream (64 batched proposals per step over a set of chunks) produced by the
    MC3 driver: initial-migration model, 64 chains, one Likelihood.batch per step."""
    model = models.IsolationMigrationModel(10, 10)
    theta0 = (0.001, 0.001, 1000.0, 0.4, 200.0)           # scripts/initial-migration-model.py:61-67 defaults
    chunks = make_data(model, theta0, 8, 20_000, seed=31)
    fw = [Forwarder.from_array(c, 3) for c in chunks]

    def run(likelihood):
        rng = np.random.default_rng(77)
        priors = [mcmc.LogNormPrior(np.log(theta0[0]), rng=rng), mcmc.LogNormPrior(np.log(theta0[1]), rng=rng),
                  mcmc.LogNormPrior(np.log(theta0[2]), rng=rng), mcmc.LogNormPrior(np.log(theta0[3]), rng=rng),
                  mcmc.ExpLogNormPrior(theta0[4], rng=rng)]
        rec = Recording(likelihood)
        chain = mcmc.MC3(priors, rec, no_chains=64, thinning=4, switching=2, temperature_scale=1.5, rng=rng)
        samples = [chain.sample() for _ in range(2)]
        state = [(c.current_theta.copy(), c.current_posterior) for c in chain.chains]
        return rec.batches, samples, state

    dev_b, dev_s, dev_state = run(Likelihood(model, fw))
    assert "k_zpropagate" in _capi.last_plan()["kernels"]
    ref_b, ref_s, ref_state = run(OracleLikelihood(oracle, model, chunks))
    assert len(dev_b) == 1 + 2 * 4 and all(b.shape == (64,) for b in dev_b)
    same_batches(dev_b, ref_b)
    for (ta, pa), (tb, pb) in zip(dev_state, ref_state):      # every chain ended in the same place
        assert np.array_equal(ta, tb) and rel_err(pa, pb) < TOL
    for a, b in zip(dev_s, ref_s):
        assert np.array_equal(a[0], b[0])


def test_mc3_large_model_trajectory(oracle):
    """The same driver on the 150-state model (the GEMM / mat-vec chain kernels): 8 chains, short chunks."""
    model = models.IsolationMigrationModel(75, 75)
    theta0 = (0.001, 0.001, 1000.0, 0.4, 200.0)
    chunks = make_data(model, theta0, 4, 6_000, seed=41)
    fw = [Forwarder.from_array(c, 3) for c in chunks]

    def run(likelihood):
        rng = np.random.default_rng(5)
        priors = [mcmc.LogNormPrior(np.log(t), rng=rng) for t in theta0[:4]] + [mcmc.ExpLogNormPrior(theta0[4], rng=rng)]
        rec = Recording(likelihood)
        chain = mcmc.MC3(priors, rec, no_chains=8, thinning=2, switching=1, temperature_scale=2.0, rng=rng)
        chain.sample()
        return rec.batches, [c.current_theta.copy() for c in chain.chains]

    dev_b, dev_t = run(Likelihood(model, fw))
    ref_b, ref_t = run(OracleLikelihood(oracle, model, chunks))
    same_batches(dev_b, ref_b)
    assert all(np.array_equal(a, b) for a, b in zip(dev_t, ref_t))


def _transform(position):
    """[0,1]^3 -> (split_time, coal_rate, recomb_rate), the default ranges of scripts/heuristic-optimiser.py:207-219."""
    p = np.asarray(position, dtype=np.float64)
    return np.array([p[0] * 0.004, p[1] * 2000.0, p[2] * 0.8])


def test_particle_swarm_trajectory(oracle):
    model = models.IsolationModel(10)
    chunks = make_data(model, (0.001, 1000.0, 0.4), 3, 30_000, seed=51)
    fw = [Forwarder.from_array(c, 3) for c in chunks]

    def run(likelihood):
        rec = Recording(likelihood)
        opt = pso.Optimiser(seed=11)
        opt.particle_count, opt.max_iterations = 16, 5
        ctx = opt.maximise(lambda positions: rec.batch([_transform(x) for x in positions]), 3)
        return rec.batches, ctx

    dev_b, dev = run(Likelihood(model, fw))
    ref_b, ref = run(OracleLikelihood(oracle, model, chunks))
    same_batches(dev_b, ref_b)
    assert len(dev_b) == 5 and dev.evaluations == 16 * 5
    assert np.array_equal(dev.positions, ref.positions) and np.array_equal(dev.best_positions, ref.best_positions)
    assert np.array_equal(dev.swarm_best_position, ref.swarm_best_position)
    assert rel_err(dev.swarm_best_fitness, ref.swarm_best_fitness) < TOL and np.isfinite(dev.swarm_best_fitness)


def test_genetic_algorithm_trajectory(oracle):
    """Each generation's offspring in one Likelihood.batch (genetic_algorithm.py:818-831)."""
    model = models.IsolationModel(10)
    chunks = make_data(model, (0.001, 1000.0, 0.4), 3, 30_000, seed=61)
    fw = [Forwarder.from_array(c, 3) for c in chunks]

    def run(likelihood):
        rec = Recording(likelihood)
        opt = ga.Optimiser(random.Random(2024))
        opt.population_size, opt.max_generations, opt.elite_count = 16, 4, 2
        ctx = opt.maximise(lambda genomes: rec.batch([_transform(g) for g in genomes]), 3)
        return rec.batches, ctx

    dev_b, dev = run(Likelihood(model, fw))
    ref_b, ref = run(OracleLikelihood(oracle, model, chunks))
    same_batches(dev_b, ref_b)
    assert [len(b) for b in dev_b] == [16, 14, 14, 14]
    assert [i.genome for i in dev.population] == [i.genome for i in ref.population]
    assert [i.genome for i in dev.hall_of_fame] == [i.genome for i in ref.hall_of_fame]
    assert np.isfinite(dev.hall_of_fame[0].fitness)
