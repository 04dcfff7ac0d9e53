"""Batched counterparts of the reference's MCMC drivers (src/IMCoalHMM/mcmc.py).

The reference runs Metropolis-coupled MCMC with one OS process per chain (mcmc.py:99-145), each owning its
own Forwarders, so that k likelihood evaluations are in flight per round (mcmc.py:165-174).  On one GPU the
same k evaluations are ONE ``Likelihood.batch`` call: the chains advance in lock step inside a single
process, which is the evaluation stream BASELINE config 5 ("64 batched proposals/step") describes.
Chains are independent between swap proposals, so lock-stepping them changes nothing statistically.

Same class names and argument meaning as the reference; an optional ``rng`` (numpy Generator) makes runs
reproducible.
"""
from math import exp, log

import numpy as np
from scipy.stats import expon, norm


class LogNormPrior(object):
    """Log-normal prior, random-walk proposal in log space (mcmc.py:16-36)."""

    def __init__(self, log_mean, proposal_sd=None, rng=None):
        self.log_mean = log_mean
        self.proposal_sd = 0.1 if proposal_sd is None else proposal_sd
        self.rng = rng if rng is not None else np.random.default_rng()

    def pdf(self, x):
        return norm.pdf(log(x), loc=self.log_mean)

    def sample(self):
        return exp(self.rng.normal(self.log_mean, 1.0))

    def proposal(self, x):
        return exp(self.rng.normal(log(x), self.proposal_sd))


class ExpLogNormPrior(object):
    """Exponential prior, random-walk proposal in log space (mcmc.py:39-57)."""

    def __init__(self, mean, proposal_sd=None, rng=None):
        self.mean = mean
        self.proposal_sd = 0.1 if proposal_sd is None else proposal_sd
        self.rng = rng if rng is not None else np.random.default_rng()

    def pdf(self, x):
        return expon.pdf(x, scale=self.mean)

    def sample(self):
        return self.rng.exponential(self.mean)

    def proposal(self, x):
        return exp(self.rng.normal(log(x), self.proposal_sd))


def _log_prior(priors, theta):
    total = 0.0
    for prior, x in zip(priors, theta):
        pdf = prior.pdf(x)
        if pdf <= 0.0:                      # mcmc.py:74-77
            return -float("inf")
        total += log(pdf)
    return total


def _batch(log_likelihood, thetas):
    """Evaluate many parameter points: one device pass if the likelihood offers ``batch``."""
    if hasattr(log_likelihood, "batch"):
        return np.asarray(log_likelihood.batch(thetas), dtype=np.float64)
    return np.array([log_likelihood(t) for t in thetas], dtype=np.float64)


class _ChainState(object):
    __slots__ = ("current_theta", "current_prior", "current_likelihood", "current_posterior")


class MCMC(object):
    """A single Metropolis chain (mcmc.py:60-96)."""

    def __init__(self, priors, log_likelihood, thinning, rng=None):
        self.priors = priors
        self.log_likelihood = log_likelihood
        self.thinning = thinning
        self.rng = rng if rng is not None else np.random.default_rng()
        self.current_theta = np.array([p.sample() for p in self.priors])
        self.current_prior = _log_prior(self.priors, self.current_theta)
        self.current_likelihood = self.log_likelihood(self.current_theta)
        self.current_posterior = self.current_prior + self.current_likelihood

    def log_prior(self, theta):
        return _log_prior(self.priors, theta)

    def step(self, temperature=1.0):
        new_theta = np.array([p.proposal(x) for p, x in zip(self.priors, self.current_theta)])
        new_prior = self.log_prior(new_theta)
        new_likelihood = self.log_likelihood(new_theta)
        new_posterior = new_prior + new_likelihood
        if new_posterior > self.current_posterior or \
                self.rng.random() < exp(new_posterior / temperature - self.current_posterior / temperature):
            self.current_theta, self.current_prior = new_theta, new_prior
            self.current_likelihood, self.current_posterior = new_likelihood, new_posterior

    def sample(self, temperature=1.0):
        for _ in range(self.thinning):
            self.step(temperature)
        return self.current_theta, self.current_prior, self.current_likelihood, self.current_posterior


class MC3(object):
    """Metropolis-coupled MCMC (mcmc.py:148-193) with all chains in one process and one batched likelihood
    evaluation per step.

    ``MC3(priors, log_likelihood, no_chains, thinning, switching, temperature_scale)``; use
    ``MC3.from_files(priors, input_files, model, ...)`` for the reference's argument list.
    """

    def __init__(self, priors, log_likelihood, no_chains, thinning, switching, temperature_scale, rng=None):
        self.priors = priors
        self.log_likelihood = log_likelihood
        self.no_chains = no_chains
        self.thinning = thinning
        self.switching = switching
        self.temperature_scale = temperature_scale
        self.rng = rng if rng is not None else np.random.default_rng()
        thetas = [np.array([p.sample() for p in priors]) for _ in range(no_chains)]
        liks = _batch(log_likelihood, thetas)
        self.chains = []
        for theta, lik in zip(thetas, liks):
            c = _ChainState()
            c.current_theta = theta
            c.current_prior = _log_prior(priors, theta)
            c.current_likelihood = float(lik)
            c.current_posterior = c.current_prior + c.current_likelihood
            self.chains.append(c)

    @classmethod
    def from_files(cls, priors, input_files, model, no_chains, thinning, switching, temperature_scale, rng=None):
        """The reference's constructor arguments (mcmc.py:151): builds the Forwarders and Likelihood here."""
        from .hmm import Forwarder
        from .likelihood import Likelihood
        forwarders = [Forwarder(arg, NSYM=3) for arg in input_files]
        return cls(priors, Likelihood(model, forwarders), no_chains, thinning, switching, temperature_scale, rng)

    def chain_temperature(self, chain_no):
        return 1.0 if chain_no == 0 else chain_no * self.temperature_scale   # mcmc.py:158-162

    def _step_all(self):
        """One Metropolis step of every chain at its current temperature: k proposals, one batched evaluation."""
        proposals = [np.array([p.proposal(x) for p, x in zip(self.priors, c.current_theta)]) for c in self.chains]
        priors = [_log_prior(self.priors, t) for t in proposals]
        liks = _batch(self.log_likelihood, proposals)
        for k, c in enumerate(self.chains):
            temperature = self.chain_temperature(k)
            new_posterior = priors[k] + float(liks[k])
            if new_posterior > c.current_posterior or \
                    self.rng.random() < exp(new_posterior / temperature - c.current_posterior / temperature):
                c.current_theta, c.current_prior = proposals[k], priors[k]
                c.current_likelihood, c.current_posterior = float(liks[k]), new_posterior

    def sample(self):
        """``thinning`` steps with a chain-swap proposal after every ``switching`` steps; returns chain 0's state."""
        for _ in range(int(float(self.thinning) / self.switching)):
            for _ in range(self.switching):
                self._step_all()
            i = int(self.rng.integers(0, self.no_chains))
            j = int(self.rng.integers(0, self.no_chains))
            if i != j:
                ti, tj = self.chain_temperature(i), self.chain_temperature(j)
                ci, cj = self.chains[i], self.chains[j]
                current = ci.current_posterior / ti + cj.current_posterior / tj
                new = cj.current_posterior / ti + ci.current_posterior / tj
                if new > current or self.rng.random() < exp(new - current):
                    self.chains[i], self.chains[j] = self.chains[j], self.chains[i]
        c = self.chains[0]
        return c.current_theta, c.current_prior, c.current_likelihood, c.current_posterior

    def terminate(self):
        """Nothing to terminate (no child processes); kept for interface compatibility (mcmc.py:191-193)."""
