"""CPU tests of the batched MCMC / MC3 drivers (imcoalhmm_amd/mcmc.py) against the semantics of the
reference's src/IMCoalHMM/mcmc.py, with a synthetic likelihood (no GPU)."""
import math

import numpy as np

from imcoalhmm_amd.mcmc import MC3, MCMC, ExpLogNormPrior, LogNormPrior


class QuadLik(object):
    """log-likelihood = -0.5 * sum((log theta - mu)^2) / s^2, with call accounting."""

    def __init__(self, mu, s=0.2):
        self.mu, self.s = np.asarray(mu), s
        self.single_calls, self.batch_calls, self.batch_sizes = 0, 0, []

    def __call__(self, theta):
        self.single_calls += 1
        return float(-0.5 * np.sum((np.log(theta) - self.mu) ** 2) / self.s ** 2)

    def batch(self, thetas):
        self.batch_calls += 1
        self.batch_sizes.append(len(thetas))
        return np.array([-0.5 * np.sum((np.log(t) - self.mu) ** 2) / self.s ** 2 for t in thetas])


def test_priors_match_reference_definitions():
    rng = np.random.default_rng(1)
    p = LogNormPrior(math.log(0.01), rng=rng)
    assert p.proposal_sd == 0.1                                           # mcmc.py:25
    assert abs(p.pdf(0.01) - 1.0 / math.sqrt(2 * math.pi)) < 1e-12        # norm.pdf(log x, loc=log_mean)
    steps = np.array([math.log(p.proposal(0.01)) for _ in range(4000)])
    assert abs(steps.mean() - math.log(0.01)) < 0.01 and abs(steps.std() - 0.1) < 0.01
    e = ExpLogNormPrior(200.0, proposal_sd=0.3, rng=rng)
    assert abs(e.pdf(100.0) - math.exp(-0.5) / 200.0) < 1e-15             # expon.pdf(x, scale=mean)
    assert abs(np.mean([e.sample() for _ in range(20000)]) - 200.0) < 6.0


def test_single_chain_samples_the_target():
    rng = np.random.default_rng(2)
    mu = [math.log(0.001), math.log(1000.0)]
    priors = [LogNormPrior(mu[0], rng=rng), LogNormPrior(mu[1], rng=rng)]
    lik = QuadLik(mu)
    chain = MCMC(priors, lik, thinning=20, rng=rng)
    draws = np.array([np.log(chain.sample()[0]) for _ in range(300)])
    assert np.all(np.abs(draws[100:].mean(axis=0) - np.array(mu)) < 0.08)
    theta, prior, likelihood, posterior = chain.sample()
    assert abs(prior + likelihood - posterior) < 1e-12                    # mcmc.py:69
    assert lik.single_calls == 1 + 301 * 20 and lik.batch_calls == 0


def test_mc3_batches_one_evaluation_per_step_and_swaps():
    rng = np.random.default_rng(3)
    mu = [math.log(0.001), math.log(1000.0), math.log(0.4)]
    priors = [LogNormPrior(m, rng=rng) for m in mu]
    lik = QuadLik(mu)
    mc3 = MC3(priors, lik, no_chains=8, thinning=100, switching=10, temperature_scale=2.0, rng=rng)
    assert mc3.chain_temperature(0) == 1.0 and mc3.chain_temperature(3) == 6.0   # mcmc.py:158-162
    assert lik.batch_calls == 1 and lik.batch_sizes == [8]               # initial states in one pass
    before = [id(c) for c in mc3.chains]
    out = [mc3.sample() for _ in range(30)]
    assert lik.batch_calls == 1 + 30 * 100 and set(lik.batch_sizes) == {8} and lik.single_calls == 0
    assert [id(c) for c in mc3.chains] != before                          # chain swaps happened
    cold = np.array([np.log(o[0]) for o in out[10:]])
    assert np.all(np.abs(cold.mean(axis=0) - np.array(mu)) < 0.15)        # the cold chain samples the target
    theta, prior, likelihood, posterior = out[-1]
    assert abs(prior + likelihood - posterior) < 1e-12
    mc3.terminate()


def test_mc3_is_reproducible_with_a_seeded_rng():
    def run(seed):
        rng = np.random.default_rng(seed)
        priors = [LogNormPrior(0.0, rng=rng), ExpLogNormPrior(2.0, rng=rng)]
        m = MC3(priors, QuadLik([0.0, 0.5]), no_chains=4, thinning=20, switching=5, temperature_scale=1.5, rng=rng)
        return np.concatenate([m.sample()[0] for _ in range(5)])
    assert np.array_equal(run(7), run(7)) and not np.array_equal(run(7), run(8))
