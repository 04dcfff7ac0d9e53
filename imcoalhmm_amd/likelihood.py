"""Drop-in for IMCoalHMM.likelihood.Likelihood (reference: src/IMCoalHMM/likelihood.py:8-33).

``model`` is any object with ``valid_parameters`` and ``build_hidden_markov_model`` (the
reference's CPU-side model layer, unchanged); ``forwarders`` is one Forwarder or an iterable of
them.  All of this module's own Forwarders are evaluated in ONE library call (one propagate
launch over every chunk); foreign objects that merely have ``.forward`` are still accepted and
summed the reference way.
"""
import numpy as np

from . import hmm


class Likelihood(object):
    """Combining model and data (likelihood.py:8-33)."""

    def __init__(self, model, forwarders):
        self.model = model
        if hasattr(forwarders, '__iter__'):
            self.forwarders = list(forwarders)
        else:
            self.forwarders = [forwarders]

    def _split(self):
        ours = [f for f in self.forwarders if isinstance(f, hmm.Forwarder)]
        return ours, len(ours) == len(self.forwarders)

    def __call__(self, *parameters):
        """Log-likelihood at one parameter point; -inf for invalid parameters (likelihood.py:29-30)."""
        if not self.model.valid_parameters(*parameters):
            return -float('inf')
        init_probs, trans_probs, emission_probs = self.model.build_hidden_markov_model(*parameters)
        ours, all_ours = self._split()
        if all_ours:
            # chunk values are summed left to right from 0.0 inside the library, like Python's sum()
            return hmm.forward_chunks([f.handle for f in ours], init_probs, trans_probs, emission_probs)
        return sum(forwarder.forward(init_probs, trans_probs, emission_probs) for forwarder in self.forwarders)

    def batch(self, thetas):
        """Evaluate many parameter points in one device pass -> float64[len(thetas)].

        Invalid points get -inf without touching the device, exactly as ``__call__`` would return.
        """
        thetas = [np.asarray(t, dtype=np.float64) for t in thetas]
        out = np.full(len(thetas), -np.inf, dtype=np.float64)
        valid = [k for k, t in enumerate(thetas) if self.model.valid_parameters(t)]
        if not valid:
            return out
        hmms = [self.model.build_hidden_markov_model(thetas[k]) for k in valid]
        ours, all_ours = self._split()
        if all_ours:
            pis = np.stack([np.asarray(h[0], dtype=np.float64).reshape(-1) for h in hmms])
            Ts = np.stack([np.asarray(h[1], dtype=np.float64) for h in hmms])
            Es = np.stack([np.asarray(h[2], dtype=np.float64) for h in hmms])
            vals = hmm.forward_chunks_batch([f.handle for f in ours], pis, Ts, Es)
        else:
            vals = [sum(f.forward(*h) for f in self.forwarders) for h in hmms]
        for k, v in zip(valid, vals):
            out[k] = v
        return out
