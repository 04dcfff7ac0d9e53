"""Drop-in for IMCoalHMM.hmm.Forwarder (reference: src/IMCoalHMM/hmm.py:10-21) on MI355X.

Same constructor and ``forward`` signature as the reference class; the observation sequence is
parsed by the C library, copied to HBM once and stays resident (the role that
``ziphmm.preprocess_raw_observations`` plays at hmm.py:16), and ``forward`` runs the HIP
parallel-in-time forward (the role of ``ziphmm.zip_forward`` at hmm.py:20-21).
"""
import ctypes
import os

import numpy as np

from . import _capi


def _ready(a):
    return type(a) is np.ndarray and a.dtype == np.float64 and a.flags.c_contiguous


def _params(init_probs, trans_probs, emission_probs):
    if _ready(init_probs) and _ready(trans_probs) and _ready(emission_probs) and init_probs.ndim == 1:
        pi, T, E = init_probs, trans_probs, emission_probs          # (the common case: nothing to convert)
    else:
        pi = _capi.as_f64(np.asarray(init_probs)).reshape(-1)
        T = _capi.as_f64(np.asarray(trans_probs))
        E = _capi.as_f64(np.asarray(emission_probs))
    n = pi.shape[0]
    if T.shape != (n, n):
        raise ValueError("trans_probs must be (%d,%d), got %r" % (n, n, T.shape))
    if E.ndim != 2 or E.shape[0] != n:
        raise ValueError("emission_probs must be (%d,NSYM), got %r" % (n, E.shape))
    return pi, T, E


def _batch_params(pis, Ts, Es):
    pis = _capi.as_f64(np.asarray(pis))
    Ts = _capi.as_f64(np.asarray(Ts))
    Es = _capi.as_f64(np.asarray(Es))
    if pis.ndim != 2 or Ts.ndim != 3 or Es.ndim != 3:
        raise ValueError("batched parameters must be pis[B,N], Ts[B,N,N], Es[B,N,S]")
    B, n = pis.shape
    if Ts.shape != (B, n, n) or Es.shape[:2] != (B, n):
        raise ValueError("inconsistent batched parameter shapes %r %r %r" % (pis.shape, Ts.shape, Es.shape))
    return pis, Ts, Es


class HandleArray(object):
    """The ``imc_obs*`` array of a fixed set of chunks, built once: with hundreds of files the list comprehension and the
    ctypes array per evaluation cost as much as a small device pass (``Likelihood`` keeps one)."""

    def __init__(self, handles):
        handles = list(handles)
        self.count = len(handles)
        self.array = _capi.handle_array(handles)


def _handles(handles):
    if isinstance(handles, HandleArray):
        return handles.array, handles.count
    if isinstance(handles, ctypes.Array):
        return handles, len(handles)
    return _capi.handle_array(handles), len(handles)


def forward_chunks(handles, pi, T, E):
    """Sum of chunk log-likelihoods for one parameter set (likelihood.py:33)."""
    pi, T, E = _params(pi, T, E)
    harr, count = _handles(handles)
    return _capi.forward1(harr, count, pi, T, E)


def forward_chunks_batch(handles, pis, Ts, Es, per_chunk=False):
    pis, Ts, Es = _batch_params(pis, Ts, Es)
    B, n = pis.shape
    S = Es.shape[2]
    L = _capi.lib()
    harr, count = _handles(handles)
    if per_chunk:
        out = np.zeros((B, count), dtype=np.float64)
        _capi.check(L.imc_forward_batch_per_chunk(harr, count, B, n, S,
                                                  _capi.dptr(pis), _capi.dptr(Ts), _capi.dptr(Es), _capi.dptr(out)))
    else:
        out = np.zeros(B, dtype=np.float64)
        _capi.check(L.imc_forward_batch(harr, count, B, n, S,
                                        _capi.dptr(pis), _capi.dptr(Ts), _capi.dptr(Es), _capi.dptr(out)))
    return out


def recompress(forwarders):
    """Retrain the shared pair dictionary on ALL the given Forwarders' alignments and re-encode them with it
    (``imc_obs_recompress``): the counterpart of ziphmm's per-file preprocessing (hmm.py:16) for a data set that comes as
    many chunks - the dictionary the first chunk alone trained is small.  One-time cost, results change by
    re-association only.  No-op for fewer than two Forwarders."""
    ours = [f for f in forwarders if isinstance(f, Forwarder)]
    if len(ours) < 2:
        return
    _capi.check(_capi.lib().imc_obs_recompress(_capi.handle_array([f.handle for f in ours]), len(ours)))


def forward_states(handles, pis, Ts, Es, as_operator):
    """State of every chunk instead of its log-likelihood (``imc_forward_state``).

    ``as_operator=False``: forward vectors started from pi -> ``(values[B, n, N], exponents[B, n])``;
    ``as_operator=True``: exact transfer operators -> ``(values[B, n, N, N], exponents[B, n, N])`` with one
    exponent per column.  True value = ``values * 2**exponents``.
    """
    pis, Ts, Es = _batch_params(pis, Ts, Es)
    B, n = pis.shape
    k = len(handles)
    state = np.zeros((B, k, n, n) if as_operator else (B, k, n), dtype=np.float64)
    exps = np.zeros((B, k, n) if as_operator else (B, k), dtype=np.int32)
    _capi.check(_capi.lib().imc_forward_state(
        _capi.handle_array(handles), k, 1 if as_operator else 0, B, n, Es.shape[2], _capi.dptr(pis),
        _capi.dptr(Ts), _capi.dptr(Es), _capi.dptr(state), exps.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
    return state, exps


def combine_states(vector, vector_exp, operators, operator_exps):
    """log sum_i (P_k ... P_1 a)_i for one parameter set: ``vector`` a (N,) with scalar exponent, then the
    operators ``(k, N, N)`` with per-column exponents ``(k, N)`` in slice order.  Exact power-of-two rescaling."""
    a = np.asarray(vector, dtype=np.float64)
    e = int(vector_exp)
    for P, pe in zip(operators, operator_exps):
        pe = np.asarray(pe, dtype=np.int64)
        top = int(pe.max())
        a = np.asarray(P, dtype=np.float64) @ np.ldexp(a, (pe - top).astype(np.int32))
        e += top
        m = a.max()
        if m > 0 and np.isfinite(m):
            shift = int(np.frexp(m)[1])
            a = np.ldexp(a, -shift)
            e += shift
    total = a.sum()
    return float(np.log(total) + e * np.log(2.0)) if total > 0 else (float('-inf') if total == 0 else float('nan'))


class Forwarder(object):
    """``Forwarder(input_filename, NSYM)`` - same surface as the reference class (hmm.py:10-21).

    ``input_filename`` is a text file of whitespace-separated symbols as written by
    scripts/prepare-alignments.py.  Extra, additive constructors: ``Forwarder.from_array``.
    """

    def __init__(self, input_filename, NSYM):
        self.NSYM = int(NSYM)
        self._h = None
        h = ctypes.c_void_p()
        _capi.check(_capi.lib().imc_obs_create_from_text(os.fsencode(input_filename), self.NSYM, ctypes.byref(h)))
        self._h = h.value
        self._pid = os.getpid()

    @classmethod
    def from_array(cls, obs, NSYM):
        """Build from an in-memory symbol array (uint8 or any integer dtype)."""
        self = cls.__new__(cls)
        self.NSYM = int(NSYM)
        self._h = None
        h = ctypes.c_void_p()
        obs = np.asarray(obs)
        L = _capi.lib()
        if obs.dtype == np.uint8:
            a = np.ascontiguousarray(obs)
            _capi.check(L.imc_obs_create(a.ctypes.data_as(_capi._u8p), a.size, self.NSYM, ctypes.byref(h)))
        else:
            a = np.ascontiguousarray(obs, dtype=np.int32)     # the reference's own dtype (hmm.py:14)
            _capi.check(L.imc_obs_create_i32(a.ctypes.data_as(_capi._i32p), a.size, self.NSYM, ctypes.byref(h)))
        self._h = h.value
        self._pid = os.getpid()
        return self

    def __len__(self):
        return int(_capi.lib().imc_obs_length(self._h))

    def compressed_length(self, alphabet_limit=256):
        """(tokens, alphabet) of the pair-compressed stream - the (new_obs, new_nsyms) of hmm.py:16."""
        used = ctypes.c_int(0)
        n = _capi.lib().imc_obs_compressed_length(self._h, int(alphabet_limit), ctypes.byref(used))
        return int(n), int(used.value)

    # The attributes the reference's Forwarder keeps from ziphmm.preprocess_raw_observations (hmm.py:15-16).  Nothing
    # outside hmm.py reads them; here they are views of the library's dictionary at its deepest level.
    @property
    def new_nsyms(self):
        return self.compressed_length(1 << 30)[1]

    @property
    def sym2pair(self):
        """{new symbol: (left, right)} for every merged token (the pair it replaces, in stream order)."""
        L = _capi.lib()
        used = ctypes.c_int(0)
        _capi.check(L.imc_obs_dictionary(self._h, 1 << 30, None, None, 0, ctypes.byref(used)))
        k = used.value - self.NSYM
        left, right = np.zeros(max(k, 1), dtype=np.uint16), np.zeros(max(k, 1), dtype=np.uint16)
        u16 = ctypes.POINTER(ctypes.c_uint16)
        _capi.check(L.imc_obs_dictionary(self._h, 1 << 30, left.ctypes.data_as(u16), right.ctypes.data_as(u16), k,
                                         ctypes.byref(used)))
        return {self.NSYM + i: (int(left[i]), int(right[i])) for i in range(k)}

    @property
    def new_obs(self):
        """The compressed observation sequence (int32 token ids), copied back from the device."""
        L = _capi.lib()
        n = ctypes.c_size_t(0)
        _capi.check(L.imc_obs_tokens(self._h, 1 << 30, None, 0, ctypes.byref(n), None))
        out = np.zeros(max(n.value, 1), dtype=np.uint16)
        _capi.check(L.imc_obs_tokens(self._h, 1 << 30, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), n.value,
                                     ctypes.byref(n), None))
        return out[:n.value].astype(np.int32)

    @property
    def handle(self):
        return self._h

    def forward(self, init_probs, trans_probs, emission_probs):
        """log P(observations | pi, T, E) - reference: hmm.py:19-21."""
        return forward_chunks([self._h], init_probs, trans_probs, emission_probs)

    def forward_batch(self, pis, Ts, Es):
        """B parameter sets in one device pass: returns float64[B]."""
        return forward_chunks_batch([self._h], pis, Ts, Es)

    def close(self):
        if getattr(self, "_h", None) and self._pid == os.getpid():
            try:
                _capi.lib().imc_obs_free(self._h)
            except Exception:
                pass
        self._h = None

    def __del__(self):
        self.close()
