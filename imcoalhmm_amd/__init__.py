"""imcoalhmm_amd - MI355X-native HMM forward log-likelihood engine for IMCoalHMM.

Host-side mirror of the reference's interface for the one accelerated path:
``Forwarder`` (IMCoalHMM.hmm.Forwarder) and ``Likelihood`` (IMCoalHMM.likelihood.Likelihood),
both backed by the C-ABI HIP library ``libimcoal_fwd.so`` (include/imcoal_fwd.h).
"""
from .hmm import Forwarder
from .likelihood import Likelihood, maximum_likelihood_estimate

__all__ = ["Forwarder", "Likelihood", "maximum_likelihood_estimate"]
__version__ = "0.1.0"
