"""Drop-in for IMCoalHMM.likelihood.Likelihood (reference: src/IMCoalHMM/likelihood.py:8-33).

``model`` is any object with ``valid_parameters`` and ``build_hidden_markov_model`` (the
reference's CPU-side model layer, unchanged); ``forwarders`` is one Forwarder or an iterable of
them.  All of this module's own Forwarders are evaluated in ONE library call (one propagate
launch over every chunk); foreign objects that merely have ``.forward`` are still accepted and
summed the reference way.
"""
import numpy as np

from . import hmm


def build_hmms(model, thetas):
    """Stacked float64 ``(pis[B,N], Ts[B,N,N], Es[B,N,S])`` for a list of parameter points.

    Uses ``model.build_batch`` when the model has one (imcoalhmm_amd.models), otherwise one
    ``build_hidden_markov_model`` call per point (the reference's model classes, model.py:44-49).
    """
    if hasattr(model, "build_batch"):
        pis, Ts, Es = model.build_batch(thetas)
    else:
        hmms = [model.build_hidden_markov_model(t) for t in thetas]
        pis = np.stack([np.asarray(h[0], dtype=np.float64).reshape(-1) for h in hmms])
        Ts = np.stack([np.asarray(h[1], dtype=np.float64) for h in hmms])
        Es = np.stack([np.asarray(h[2], dtype=np.float64) for h in hmms])
    return (np.ascontiguousarray(pis, dtype=np.float64), np.ascontiguousarray(Ts, dtype=np.float64),
            np.ascontiguousarray(Es, dtype=np.float64))


class Likelihood(object):
    """Combining model and data (likelihood.py:8-33)."""

    def __init__(self, model, forwarders, recompress=True):
        self.model = model
        if hasattr(forwarders, '__iter__'):
            self.forwarders = list(forwarders)
        else:
            self.forwarders = [forwarders]
        if recompress:      # several chunks: one pair dictionary trained on all of them (hmm.recompress)
            hmm.recompress(self.forwarders)

    def _split(self):
        ours = [f for f in self.forwarders if isinstance(f, hmm.Forwarder)]
        return ours, len(ours) == len(self.forwarders)

    def _handle_array(self):
        """The chunk handles as one array, built at the first evaluation (rebuilt if the list of forwarders is replaced or
        changes length; its members are not expected to change behind it): None if some forwarder is not ours."""
        cached = getattr(self, "_harr", None)
        if cached is not None and cached[0] is self.forwarders and cached[1] == len(self.forwarders):
            return cached[2]
        ours, all_ours = self._split()
        harr = hmm.HandleArray(f.handle for f in ours) if all_ours else None
        self._harr = (self.forwarders, len(self.forwarders), harr)
        return harr

    def __call__(self, *parameters):
        """Log-likelihood at one parameter point; -inf for invalid parameters (likelihood.py:29-30)."""
        if not self.model.valid_parameters(*parameters):
            return -float('inf')
        init_probs, trans_probs, emission_probs = self.model.build_hidden_markov_model(*parameters)
        harr = self._handle_array()
        if harr is not None:
            # chunk values are summed left to right from 0.0 inside the library, like Python's sum()
            return hmm.forward_chunks(harr, init_probs, trans_probs, emission_probs)
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
        pis, Ts, Es = build_hmms(self.model, [thetas[k] for k in valid])
        harr = self._handle_array()
        if harr is not None:
            vals = hmm.forward_chunks_batch(harr, pis, Ts, Es)
        else:
            vals = [sum(f.forward(pis[b], Ts[b], Es[b]) for f in self.forwarders) for b in range(len(valid))]
        for k, v in zip(valid, vals):
            out[k] = v
        return out


def maximum_likelihood_estimate(log_likelihood, initial_parameters, optimizer_method="Nelder-Mead",
                                log_file=None, log_param_transform=lambda x: x):
    """Maximise ``log_likelihood`` from ``initial_parameters`` with a scipy optimiser (likelihood.py:36-87).

    Same arguments and return value (the arg-max as an ndarray) as the reference: bounded methods get
    ``(0, None)`` bounds per parameter, every accepted iterate is written tab-separated to
    ``log_file`` after ``log_param_transform``.
    """
    import scipy.optimize

    callback = None
    if log_file:
        def callback(parameters):
            log_file.write('\t'.join(str(p) for p in log_param_transform(parameters)) + '\n')

    def negated(parameters):
        return -log_likelihood(parameters)

    kwargs = {}
    if optimizer_method in ('Anneal', 'L-BFGS-B', 'TNC', 'SLSQP'):
        kwargs['bounds'] = [(0, None)] * len(initial_parameters)
    result = scipy.optimize.minimize(fun=negated, x0=initial_parameters, method=optimizer_method,
                                     callback=callback, options={'disp': False}, **kwargs)
    return result.x
