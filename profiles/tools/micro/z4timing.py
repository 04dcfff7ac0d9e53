"""diagnostics: phase times of k_zpropagate4 on BASELINE config[1] data (needs a library built with -DIMC_Z4_TIMING)"""
import ctypes, os, sys
import numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
import bench
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
data = bench.generate([("c2", "iso20_t0", cols, 20240001)])["c2"]
from imcoalhmm_amd import _capi
from imcoalhmm_amd.hmm import Forwarder
d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
pi, T, E = d["iso20_t0_pi"], d["iso20_t0_T"], d["iso20_t0_E"]
fw = Forwarder.from_array(data, 3) if hasattr(Forwarder, "from_array") else Forwarder(data, 3)
for _ in range(4):
    v = fw.forward(pi, T, E)
print("loglik", v)
lib = _capi.lib() if callable(getattr(_capi, "lib", None)) else _capi.LIB
fn = lib.imc_debug_z4_timing
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros(3 * 1024 * 8, dtype=np.int64)
assert fn(buf.ctypes.data, buf.size) == 0
allb = buf.reshape(3, 1024, 8)
sel = allb[0][:, 0] != 0
t = allb[0][sel]
wv = allb[1][sel]
fl = allb[2][sel]
print("workgroups", len(t))
rel = (t[:, :6] - t[:, :1].min()) / 100.0       # us since the first workgroup started
names = ["start", "lds ready", "first block done", "full blocks done", "tail blocks done", "fold done"]
for k, n in enumerate(names):
    print("%-18s min %7.1f  median %7.1f  max %7.1f us" % (n, rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
dur = np.diff(t[:, :6], axis=1) / 100.0
for k in range(5):
    print("phase %d (%s -> %s): median %6.1f  max %6.1f us" % (k, names[k], names[k + 1], np.median(dur[:, k]), dur[:, k].max()))
nfull = t[:, 7] // 1000000; maxlen = t[:, 7] % 1000000
print("nfull", np.bincount(nfull)[:12], "maxlen", np.unique(maxlen)[:10])

skew = (wv.max(axis=1) - wv.min(axis=1)) / 100.0
print("scan end skew between the 8 wavefronts of a workgroup: median %.1f max %.1f us" % (np.median(skew), skew.max()))
print("slowest wavefront after wavefront 0: median %.1f us" % np.median((wv.max(axis=1) - wv[:, 0]) / 100.0))
lv = np.concatenate([wv.max(axis=1)[:, None], fl[:, :5]], axis=1)
print("fold levels (from the slowest wavefront's scan end): median us", np.round(np.median(np.diff(lv, axis=1), axis=0) / 100.0, 1))
base = t[:, :1].min()
print("scan end per wavefront index (median us):", np.round(np.median((wv - base) / 100.0, axis=0), 1))
print("scan end per wavefront index (max us):   ", np.round(np.max((wv - base) / 100.0, axis=0), 1))
