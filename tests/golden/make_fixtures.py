#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (run once, in the build container).

What it does
------------
1. (theta -> pi, T, E) fixtures.  The reference's model layer (CPU, numpy/scipy; SURVEY.md
   section 2 rows 6-13) is Python 2.  A scratch copy is made under /tmp, converted with the stock
   ``lib2to3`` tool, and imported from there (never from the repo, never committed).  Only
   ``IsolationModel`` and ``IsolationMigrationModel`` are imported - they stay CPU-side in the build
   and *feed* the hot path (reference: src/IMCoalHMM/model.py:44-49).  ``IMCoalHMM.hmm`` is NOT
   imported (it needs the absent ``ziphmm`` module).
2. Observation fixtures: the six species pairs of the reference's only shipped alignment
   (examples/example_data.fa) encoded by the rule at scripts/prepare-alignments.py:92-105
   (2 = either base not in ACGT, 0 = equal, 1 = different), stored as uint8 arrays.
3. Golden log-likelihoods: computed by the C oracle (oracle/forward_oracle.c, scaled forward) and
   cross-checked here against the numpy textbook forward (oracle/forward_numpy.py) and the
   long-double variant; frozen into loglik_golden.json.  NOTE: no reference test pins a forward
   value and ``ziphmm`` is not installable offline, so these are "parity unpinned" against ziphmm
   itself - they pin the mathematical definition, which zipHMM re-associates.

Outputs (all small, data only):  hmm_params.npz, example_pairs.npz, loglik_golden.json
"""
import itertools
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
SCRATCH = "/tmp/imc3"


def import_reference_models():
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    os.makedirs(SCRATCH)
    shutil.copytree(os.path.join(REF, "src", "IMCoalHMM"), os.path.join(SCRATCH, "IMCoalHMM"))
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", "IMCoalHMM"], cwd=SCRATCH,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    import scipy
    scipy.matrix = np.matrix          # alias removed from current SciPy (CTMC.py:5)
    sys.path.insert(0, SCRATCH)
    from IMCoalHMM.isolation_model import IsolationModel
    from IMCoalHMM.isolation_with_migration_model import IsolationMigrationModel
    return IsolationModel, IsolationMigrationModel


def build_params():
    IsolationModel, IsolationMigrationModel = import_reference_models()
    out = {}
    # defaults of scripts/isolation-model.py:54-58,78-80: split 1e-3, theta 1e-3 -> coal 2000?  The
    # survey's reference point is coal_rate=1000 (theta=2e-3); we keep that and add perturbed points.
    iso_thetas = [
        (0.001, 1000.0, 0.4),
        (0.0005, 1500.0, 0.8),
        (0.002, 600.0, 0.1),
        (0.001, 2000.0, 0.4),   # literal script default: 1/(1e-3/2)
    ]
    for n in (10, 20):
        m = IsolationModel(n)
        for k, th in enumerate(iso_thetas):
            pi, T, E = m.build_hidden_markov_model(np.array(th))
            key = "iso%d_t%d" % (n, k)
            out[key + "_theta"] = np.array(th)
            out[key + "_pi"] = np.ascontiguousarray(np.asarray(pi, dtype=np.float64)).reshape(-1)
            out[key + "_T"] = np.ascontiguousarray(np.asarray(T, dtype=np.float64))
            out[key + "_E"] = np.ascontiguousarray(np.asarray(E, dtype=np.float64))
    im_thetas = [
        (0.001, 0.001, 1000.0, 0.4, 200.0),
        (0.0007, 0.0015, 1300.0, 0.6, 120.0),
    ]
    for (a, b), ths in (((10, 10), im_thetas), ((75, 75), im_thetas[:1])):
        m = IsolationMigrationModel(a, b)
        for k, th in enumerate(ths):
            pi, T, E = m.build_hidden_markov_model(np.array(th))
            key = "im%d_t%d" % (a + b, k)
            out[key + "_theta"] = np.array(th)
            out[key + "_pi"] = np.ascontiguousarray(np.asarray(pi, dtype=np.float64)).reshape(-1)
            out[key + "_T"] = np.ascontiguousarray(np.asarray(T, dtype=np.float64))
            out[key + "_E"] = np.ascontiguousarray(np.asarray(E, dtype=np.float64))
    return out


def read_fasta(path):
    seqs, name = {}, None
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                name = line[1:].split()[0]
                seqs[name] = []
            else:
                seqs[name].append(line)
    return {k: "".join(v) for k, v in seqs.items()}


def encode_pair(s1, s2):
    """scripts/prepare-alignments.py:92-105."""
    a = np.frombuffer(s1.upper().encode(), dtype=np.uint8)
    b = np.frombuffer(s2.upper().encode(), dtype=np.uint8)
    assert a.shape == b.shape
    clean = np.zeros(256, dtype=bool)
    for ch in b"ACGT":
        clean[ch] = True
    out = np.where(a == b, 0, 1).astype(np.uint8)
    out[~(clean[a] & clean[b])] = 2
    return out


def build_pairs():
    seqs = read_fasta(os.path.join(REF, "examples", "example_data.fa"))
    names = list(seqs.keys())
    out = {}
    for n1, n2 in itertools.combinations(names, 2):
        out["%s__%s" % (n1, n2)] = encode_pair(seqs[n1], seqs[n2])
    return out


def main():
    params = build_params()
    np.savez_compressed(os.path.join(HERE, "hmm_params.npz"), **params)
    pairs = build_pairs()
    np.savez_compressed(os.path.join(HERE, "example_pairs.npz"), **pairs)
    for k, v in pairs.items():
        print(k, len(v), np.bincount(v, minlength=3))

    # golden log-likelihoods (oracle must already be built: make -C oracle)
    sys.path.insert(0, REPO)
    from oracle import oracle_lib, forward_numpy
    golden = {}
    for pname, obs in pairs.items():
        for mkey in ("iso10_t0", "iso20_t0", "iso10_t1", "iso20_t2", "im20_t0", "im150_t0"):
            pi, T, E = params[mkey + "_pi"], params[mkey + "_T"], params[mkey + "_E"]
            ll_c = oracle_lib.forward_scaled(pi, T, E, obs)
            ll_ld = oracle_lib.forward_scaled_ld(pi, T, E, obs)
            ll_np = forward_numpy.forward_loglik(pi, T, E, obs)
            ll_zip = oracle_lib.zip_forward_from_raw(pi, T, E, obs)
            rel = max(abs(ll_c - ll_ld), abs(ll_c - ll_np), abs(ll_c - ll_zip)) / abs(ll_ld)
            assert rel < 1e-12, (pname, mkey, ll_c, ll_ld, ll_np, ll_zip)
            golden["%s|%s" % (pname, mkey)] = {"loglik": ll_ld, "max_rel_spread": rel}
            print(pname, mkey, repr(ll_ld), rel)
    with open(os.path.join(HERE, "loglik_golden.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
