"""ctypes binding of oracle/liboracle_fwd.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_fwd.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    src = os.path.join(_HERE, "forward_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        for name in ("orc_forward_scaled", "orc_forward_scaled_ld"):
            f = getattr(L, name)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _u8p, ctypes.c_size_t]
        L.orc_forward_scaled_u16.restype = ctypes.c_double
        L.orc_forward_scaled_u16.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.POINTER(ctypes.c_uint16),
                                             ctypes.c_size_t]
        L.orc_zip_preprocess.restype = ctypes.c_void_p
        L.orc_zip_preprocess.argtypes = [_u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_long, ctypes.c_int]
        L.orc_zip_length.restype = ctypes.c_size_t
        L.orc_zip_length.argtypes = [ctypes.c_void_p]
        L.orc_zip_new_nsyms.restype = ctypes.c_int
        L.orc_zip_new_nsyms.argtypes = [ctypes.c_void_p]
        L.orc_zip_free.restype = None
        L.orc_zip_free.argtypes = [ctypes.c_void_p]
        L.orc_zip_forward.restype = ctypes.c_double
        L.orc_zip_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, _dp, _dp]
        L.orc_forward_chunks_mt.restype = ctypes.c_double
        L.orc_forward_chunks_mt.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp,
                                            ctypes.POINTER(_u8p), ctypes.POINTER(ctypes.c_size_t),
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, _dp]
        L.orc_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _prep(pi, T, E, obs):
    pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(-1)
    T = np.ascontiguousarray(T, dtype=np.float64)
    E = np.ascontiguousarray(E, dtype=np.float64)
    obs = np.ascontiguousarray(obs, dtype=np.uint8)
    N, S = E.shape
    assert T.shape == (N, N) and pi.shape == (N,)
    assert obs.size == 0 or int(obs.max()) < S
    return pi, T, E, obs, N, S


def _p(a):
    return a.ctypes.data_as(_dp)


def forward_scaled(pi, T, E, obs):
    if np.asarray(E).shape[1] > 256:                     # alphabets beyond 256 symbols: 16-bit observations
        pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(-1)
        T = np.ascontiguousarray(T, dtype=np.float64)
        E = np.ascontiguousarray(E, dtype=np.float64)
        obs = np.asarray(obs)
        assert obs.size == 0 or (int(obs.min()) >= 0 and int(obs.max()) < E.shape[1])
        obs = np.ascontiguousarray(obs, dtype=np.uint16)
        return lib().orc_forward_scaled_u16(E.shape[0], E.shape[1], _p(pi), _p(T), _p(E),
                                            obs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), obs.size)
    pi, T, E, obs, N, S = _prep(pi, T, E, obs)
    return lib().orc_forward_scaled(N, S, _p(pi), _p(T), _p(E), obs.ctypes.data_as(_u8p), obs.size)


def forward_scaled_ld(pi, T, E, obs):
    pi, T, E, obs, N, S = _prep(pi, T, E, obs)
    return lib().orc_forward_scaled_ld(N, S, _p(pi), _p(T), _p(E), obs.ctypes.data_as(_u8p), obs.size)


class Zip:
    """Compressed observation sequence (the preprocess_raw_observations analogue)."""

    def __init__(self, obs, nsym, min_count=64, max_new=256):
        obs = np.ascontiguousarray(obs, dtype=np.uint8)
        self.nsym = nsym
        self.h = lib().orc_zip_preprocess(obs.ctypes.data_as(_u8p), obs.size, nsym, min_count, max_new)
        self.length = lib().orc_zip_length(self.h)
        self.new_nsyms = lib().orc_zip_new_nsyms(self.h)

    def forward(self, pi, T, E):
        pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(-1)
        T = np.ascontiguousarray(T, dtype=np.float64)
        E = np.ascontiguousarray(E, dtype=np.float64)
        assert E.shape[1] == self.nsym
        return lib().orc_zip_forward(self.h, len(pi), _p(pi), _p(T), _p(E))

    def __del__(self):
        try:
            if self.h:
                lib().orc_zip_free(self.h)
                self.h = None
        except Exception:
            pass


def zip_forward_from_raw(pi, T, E, obs, min_count=64, max_new=256):
    E = np.asarray(E)
    return Zip(obs, E.shape[1], min_count, max_new).forward(pi, T, E)


def forward_chunks_mt(pi, T, E, chunks, threads=0, zips=None):
    """Sum of per-chunk log-likelihoods on `threads` host threads; returns (total, per_chunk)."""
    pi = np.ascontiguousarray(pi, dtype=np.float64).reshape(-1)
    T = np.ascontiguousarray(T, dtype=np.float64)
    E = np.ascontiguousarray(E, dtype=np.float64)
    N, S = E.shape
    chunks = [np.ascontiguousarray(c, dtype=np.uint8) for c in chunks]
    n = len(chunks)
    ptrs = (_u8p * n)(*[c.ctypes.data_as(_u8p) for c in chunks])
    lens = (ctypes.c_size_t * n)(*[c.size for c in chunks])
    zarr = None
    if zips is not None:
        zarr = (ctypes.c_void_p * n)(*[z.h for z in zips])
    per = np.zeros(n, dtype=np.float64)
    tot = lib().orc_forward_chunks_mt(N, S, _p(pi), _p(T), _p(E), ptrs, lens, zarr, n, threads, _p(per))
    return tot, per


def max_threads():
    return lib().orc_max_threads()
