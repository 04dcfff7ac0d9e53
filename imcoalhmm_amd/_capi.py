"""ctypes binding of libimcoal_fwd.so (include/imcoal_fwd.h).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is usable, every
compute call raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported
from here.)
"""
import ctypes
import threading
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# IMCOAL_FWD_LIB selects another build of the library (A/B measurements of compile-time variants); default: in-tree
LIB_PATH = os.environ.get("IMCOAL_FWD_LIB") or os.path.join(_PKG, "libimcoal_fwd.so")

IMC_OK = 0
IMC_ERR_ARG = -1
IMC_ERR_SYMBOL = -2
IMC_ERR_HIP = -3
IMC_ERR_OOM = -4
IMC_ERR_IO = -5
IMC_ERR_NODEVICE = -6

_dp = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)
_vpp = ctypes.POINTER(ctypes.c_void_p)
_u64p = ctypes.POINTER(ctypes.c_uint64)

# every symbol include/imcoal_fwd.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("imc_version", ctypes.c_char_p, []),
    ("imc_last_error", ctypes.c_char_p, []),
    ("imc_device_count", ctypes.c_int, []),
    ("imc_set_device", ctypes.c_int, [ctypes.c_int]),
    ("imc_obs_create", ctypes.c_int, [_u8p, ctypes.c_size_t, ctypes.c_int, _vpp]),
    ("imc_obs_create_i32", ctypes.c_int, [_i32p, ctypes.c_size_t, ctypes.c_int, _vpp]),
    ("imc_obs_create_from_text", ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, _vpp]),
    ("imc_read_observations", ctypes.c_int,
     [ctypes.c_char_p, ctypes.c_int, _u8p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    ("imc_write_cache", ctypes.c_int, [ctypes.c_char_p, _u8p, ctypes.c_size_t, ctypes.c_int]),
    ("imc_encode_pairwise", ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, _u8p]),
    ("imc_obs_length", ctypes.c_size_t, [ctypes.c_void_p]),
    ("imc_obs_nsym", ctypes.c_int, [ctypes.c_void_p]),
    ("imc_obs_compressed_length", ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    ("imc_obs_dictionary", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16), ctypes.c_size_t,
      ctypes.POINTER(ctypes.c_int)]),
    ("imc_obs_tokens", ctypes.c_int,
     [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint16), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
      ctypes.POINTER(ctypes.c_int)]),
    ("imc_obs_free", ctypes.c_int, [ctypes.c_void_p]),
    ("imc_forward", ctypes.c_int, [_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp]),
    ("imc_forward_batch", ctypes.c_int,
     [_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp]),
    ("imc_forward_batch_per_chunk", ctypes.c_int,
     [_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp]),
    ("imc_forward_batch_device", ctypes.c_int,
     [_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.c_void_p,
      ctypes.c_void_p]),
    ("imc_forward_state", ctypes.c_int,
     [_vpp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _dp,
      ctypes.POINTER(ctypes.c_int)]),
    ("imc_set_segment_length", ctypes.c_int, [ctypes.c_size_t]),
    ("imc_set_compression", ctypes.c_int, [ctypes.c_int]),
    ("imc_dictionary_reset", ctypes.c_int, []),
    ("imc_profile_enable", ctypes.c_int, [ctypes.c_int]),
    ("imc_profile_read", ctypes.c_int, [_dp, _dp, _u64p, _u64p]),
    ("imc_last_rank1", ctypes.c_int, [_u64p, _u64p]),
    ("imc_set_rank1_handoff", ctypes.c_int, [ctypes.c_int]),
    ("imc_set_blocked_kernel", ctypes.c_int, [ctypes.c_int]),
    ("imc_set_table_streaming", ctypes.c_int, [ctypes.c_int]),
    ("imc_obs_recompress", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    ("imc_last_plan", ctypes.c_int, [_u64p]),
    ("imc_last_kernels", ctypes.c_char_p, []),
    # include/imcoal_model.h (host-side helper, no device)
    ("imc_model_transitions", ctypes.c_int,
     [ctypes.c_int, ctypes.c_int, _i32p, _i32p, _i32p, _i32p, _i32p, ctypes.c_int, _i32p, ctypes.c_int, _i32p, _dp, _dp, _dp, _dp,
      _dp, _dp, ctypes.c_int]),
    ("imc_model_expm", ctypes.c_int, [ctypes.c_int, _dp, _dp]),
]

_lib = None


class ImcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libimcoal_fwd error %d: %s" % (code, message))
        self.code = code


def lib():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing - build it with `python -m imcoalhmm_amd.build` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, res, args in SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc == IMC_OK:
        return
    msg = lib().imc_last_error().decode("utf-8", "replace")
    if rc in (IMC_ERR_ARG, IMC_ERR_SYMBOL):
        raise ValueError("libimcoal_fwd: " + msg)
    if rc == IMC_ERR_IO:
        raise IOError("libimcoal_fwd: " + msg)
    if rc == IMC_ERR_OOM:
        raise MemoryError("libimcoal_fwd: " + msg)
    raise ImcError(rc, msg)


_fwd1 = None
_tls = threading.local()           # one output slot per thread: the library call releases the GIL


def forward1(harr, n_chunks, pi, T, E):
    """``imc_forward`` for float64 C-contiguous arrays, with as little Python in front of the call as ctypes allows
    (raw addresses instead of typed pointer objects, one reused output slot): an evaluation of BASELINE config[1] takes
    0.3 ms, and the generic wrapper spent 25 us of that."""
    global _fwd1
    if _fwd1 is None:
        proto = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
        _fwd1 = proto(("imc_forward", lib()))
    out = getattr(_tls, "out", None)
    if out is None:
        out = _tls.out = np.zeros(1, dtype=np.float64)
        _tls.addr = out.ctypes.data
    rc = _fwd1(ctypes.addressof(harr), n_chunks, pi.shape[0], E.shape[1], pi.ctypes.data, T.ctypes.data, E.ctypes.data,
               _tls.addr)
    if rc != IMC_OK:
        check(rc)
    return float(out[0])


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError("expected shape %r, got %r" % (shape, a.shape))
    return a


def dptr(a):
    return a.ctypes.data_as(_dp)


def handle_array(handles):
    arr = (ctypes.c_void_p * max(len(handles), 1))()
    for i, h in enumerate(handles):
        arr[i] = h
    return arr


def last_rank1():
    """(operator segments tested, certified rank one) by the GEMM chain's rank-one hand-off in the last call."""
    a, b = ctypes.c_uint64(), ctypes.c_uint64()
    check(lib().imc_last_rank1(ctypes.byref(a), ctypes.byref(b)))
    return int(a.value), int(b.value)


def last_plan():
    """Dict view of imc_last_plan()."""
    arr = (ctypes.c_uint64 * 8)()
    check(lib().imc_last_plan(arr))
    keys = ("segments", "vectors", "column_segment_len", "vector_columns", "token_segment_len",
            "vector_tokens", "tokens", "token_alphabet")
    d = dict(zip(keys, [int(x) for x in arr]))
    d["kernels"] = lib().imc_last_kernels().decode()
    return d
