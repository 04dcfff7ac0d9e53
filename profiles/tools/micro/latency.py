#!/usr/bin/env python3
"""Small-input latency: the reference's example pair (65,255 columns, 10 and 20 states) and the authors' data scale
(100 x 1e6 columns, 10 states), evaluated in a loop through the scalar entry point.  Run as is for wall-clock numbers
(IMC_DEBUG_HOST=1 adds the library's host-side phase times), or under `rocprofv3 --kernel-trace` for the timelines that
trace2.sh prints.  usage: latency.py [pair|files|both] [reps]"""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
from imcoalhmm_amd import Forwarder, _capi, synth          # noqa: E402
from imcoalhmm_amd.hmm import recompress                    # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
lib = _capi.lib()


def loop(tag, harr, n, key, reps):
    pi, T, E = d[key + "_pi"], d[key + "_T"], d[key + "_E"]
    for _ in range(10):
        v = _capi.forward1(harr, n, pi, T, E)
    t0 = time.perf_counter()
    for _ in range(reps):
        v = _capi.forward1(harr, n, pi, T, E)
    us = (time.perf_counter() - t0) / reps * 1e6
    print("%-40s %8.1f us per evaluation   loglik %.10f   %s" % (tag, us, v, _capi.last_plan()["kernels"]), flush=True)


if what in ("pair", "both"):
    pair = np.load(os.path.join(REPO, "tests", "golden", "example_pairs.npz"))["hg18__pantro2"]
    f = Forwarder.from_array(pair, 3)
    h = _capi.handle_array([f.handle])
    for key in ("iso10_t0", "iso20_t0"):
        loop("example pair 65255 columns, %s" % key, h, 1, key, reps)
if what in ("files", "both"):
    _capi.check(lib.imc_dictionary_reset())
    pi, T, E = d["iso10_t0_pi"], d["iso10_t0_T"], d["iso10_t0_E"]
    fw = [Forwarder.from_array(synth.sample_alignment(pi, T, E, 1_000_000, seed=20240800 + i), 3) for i in range(100)]
    recompress(fw)
    h = _capi.handle_array([f.handle for f in fw])
    loop("100 x 1e6 columns, iso10_t0", h, 100, "iso10_t0", max(reps // 10, 5))
    h1 = _capi.handle_array([fw[0].handle])
    loop("1 x 1e6 columns, iso10_t0", h1, 1, "iso10_t0", reps)
