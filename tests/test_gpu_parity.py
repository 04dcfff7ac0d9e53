"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Tolerance: north_star asks for <= 1e-9 relative on the summed log-likelihood; the engine only
re-associates fp64 arithmetic (exact power-of-two rescaling), so the tests hold it to 1e-11.
"""
import ctypes
import math
import os

import numpy as np
import pytest

from conftest import rel_err
from imcoalhmm_amd import Forwarder, Likelihood, _capi, synth
from imcoalhmm_amd.hmm import forward_chunks, forward_chunks_batch, recompress

pytestmark = pytest.mark.gpu
TOL = 1e-11


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    L = _capi.lib()
    assert L.imc_device_count() >= 1, "gpu tests need a device"
    yield
    L.imc_set_segment_length(0)


def set_seg(n):
    _capi.check(_capi.lib().imc_set_segment_length(n))


def set_zip(mode, reset=True):
    L = _capi.lib()
    _capi.check(L.imc_set_compression(mode))
    if reset:
        _capi.check(L.imc_dictionary_reset())


@pytest.fixture(params=[4, 5, 0, 2, 3, 1, 13, 15, 23, 33],
                ids=["raw-vector", "raw-blocked", "raw-auto", "token-vector", "token-blocked", "auto", "token-blocked-valu",
                     "raw-blocked-valu", "token-blocked-lds", "token-blocked-hybrid"])
def zipmode(request):
    """Run a test on the per-column kernel, on each pinned token-kernel variant and with automatic choice.  Modes 3 / 5
    pin the register-blocked kernel, by default its fp64-MFMA form (k_zpropagate3 / k_zpropagate4, chosen by cost);
    13 / 15 are the same modes with the VALU / DPP form (k_zpropagate2), 23 pins the LDS-table MFMA kernel, 33 the
    hybrid-table one wherever the dictionary has a level beyond LDS."""
    variant = {0: 4, 1: 2, 2: 3, 3: 5}[request.param // 10]
    _capi.check(_capi.lib().imc_set_blocked_kernel(variant))
    set_zip(request.param % 10)
    yield request.param % 10
    _capi.check(_capi.lib().imc_set_blocked_kernel(4))
    set_zip(1)


def compressible(n, seed, nsym=3):
    rng = np.random.default_rng(seed)
    p = np.full(nsym, 0.1 / max(nsym - 1, 1)); p[0] = 0.9
    return rng.choice(nsym, size=n, p=p / p.sum()).astype(np.uint8)


def test_config1_example_data_golden(hmm_params, example_pairs, golden_loglik, zipmode):
    """BASELINE config 1 (10 states, examples/example_data.fa) and the other golden cells, N<=20."""
    fw = {k: Forwarder.from_array(v, 3) for k, v in example_pairs.items()}
    if zipmode in (1, 2, 3):
        ntok, alpha = fw["hg18__pantro2"].compressed_length()
        assert alpha > 3 and ntok * 8 < 65255          # the reference's example alignment compresses > 8x
    for key, rec in golden_loglik.items():
        pname, mkey = key.split("|")
        pi, T, E = hmm_params(mkey)          # includes im150_t0: N=150, the intent of BASELINE configs 3/5
        got = fw[pname].forward(pi, T, E)
        assert rel_err(got, rec["loglik"]) < TOL, (key, got, rec["loglik"])


@pytest.mark.parametrize("seg", [0, 16, 48, 1000, 4096])
def test_segmentation_invariance(hmm_params, example_pairs, golden_loglik, seg, zipmode):
    """The parallel-in-time split must not change the answer (exact re-association)."""
    try:
        set_seg(seg)
        f = Forwarder.from_array(example_pairs["hg18__pantro2"], 3)
        for mkey in ("iso10_t0", "iso20_t0", "im20_t0"):
            got = f.forward(*hmm_params(mkey))
            assert rel_err(got, golden_loglik["hg18__pantro2|" + mkey]["loglik"]) < TOL, (mkey, seg)
    finally:
        set_seg(0)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 8, 10, 12, 13, 16, 17, 20, 21, 24, 25, 28, 29, 32, 33, 40, 41, 48, 50, 56, 57, 64,
                               65, 96, 97, 112, 113, 128, 129, 144, 145, 150, 160, 161, 192, 200, 225, 256])
@pytest.mark.parametrize("nsym", [3])
def test_random_hmms_all_kernel_shapes(oracle, n, nsym):
    """Every (R,G) instantiation, padded and unpadded N, with stitching forced (seg=160)."""
    pi, T, E = synth.random_hmm(n, nsym, seed=100 + n, stay=0.98)
    rng = np.random.default_rng(n)
    chunks = [rng.integers(0, nsym, size=L).astype(np.uint8) for L in (1, 15, 16, 17, 333, 1500)]
    fw = [Forwarder.from_array(c, nsym) for c in chunks]
    try:
        set_seg(160)
        got = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
    finally:
        set_seg(0)
    for c, g in zip(chunks, got):
        want = oracle.forward_scaled(pi, T, E, c)
        assert rel_err(g, want) < TOL, (n, c.size, g, want)


@pytest.mark.parametrize("nsym", [1, 2, 4, 5, 16, 65, 256])
def test_alphabet_sizes(oracle, nsym):
    pi, T, E = synth.random_hmm(9, nsym, seed=nsym, stay=0.9)
    obs = np.random.default_rng(nsym).integers(0, nsym, size=3000).astype(np.uint8)
    got = Forwarder.from_array(obs, nsym).forward(pi, T, E)
    want = oracle.forward_scaled(pi, T, E, obs)
    # nsym=1 (all-ones emissions) has loglik == 0 up to rounding: absolute floor of 1e-11
    assert abs(got - want) < TOL * max(abs(want), 1.0)


def test_large_alphabet_with_many_states(oracle):
    """S=256 with N=40: the per-column kernel's emission table needs > 64 KB of LDS (opt-in attribute)."""
    pi, T, E = synth.random_hmm(40, 256, seed=5, stay=0.9)
    obs = np.random.default_rng(5).integers(0, 256, size=2000).astype(np.uint8)
    got = Forwarder.from_array(obs, 256).forward(pi, T, E)
    assert rel_err(got, oracle.forward_scaled(pi, T, E, obs)) < TOL


def test_ragged_and_empty_chunks_sum(oracle, hmm_params, zipmode):
    """likelihood.py:33 semantics: each chunk restarts from pi, values summed left to right."""
    pi, T, E = hmm_params("iso20_t0")
    lens = [0, 1, 2, 15, 16, 17, 31, 32, 33, 1000, 4097, 65255, 0, 20000]
    chunks = [synth.sample_alignment(pi, T, E, n, seed=500 + k) for k, n in enumerate(lens)]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    want = [oracle.forward_scaled(pi, T, E, c) for c in chunks]
    for seg in (0, 64, 1024):
        try:
            set_seg(seg)
            per = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
            tot = forward_chunks([f.handle for f in fw], pi, T, E)
        finally:
            set_seg(0)
        for g, w, n in zip(per, want, lens):
            assert (g == 0.0 and w == 0.0) or rel_err(g, w) < TOL, (seg, n, g, w)
        s = 0.0
        for g in per:
            s += g
        assert tot == s                       # same left-to-right sum as Python's sum()
        assert rel_err(tot, sum(want)) < TOL
    # Likelihood over the same forwarders
    class M(object):
        def valid_parameters(self, p):
            return all(p > 0)

        def build_hidden_markov_model(self, p):
            return pi, T, E
    ll = Likelihood(M(), fw)
    assert ll(np.array([1.0])) == forward_chunks([f.handle for f in fw], pi, T, E)   # same (automatic) split
    assert rel_err(ll(np.array([1.0])), tot) < 1e-13
    assert ll(np.array([-1.0])) == -math.inf


def test_batch_of_parameter_sets(oracle, hmm_params, zipmode):
    keys = ["iso20_t0", "iso20_t1", "iso20_t2", "iso20_t3", "im20_t0", "im20_t1"]
    ps = [hmm_params(k) for k in keys]
    pis = np.stack([p[0] for p in ps]); Ts = np.stack([p[1] for p in ps]); Es = np.stack([p[2] for p in ps])
    chunks = [synth.sample_alignment(*ps[0], 30_000, seed=900 + k) for k in range(5)]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    try:
        set_seg(2048)
        got = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es)
    finally:
        set_seg(0)
    for b, k in enumerate(keys):
        want = sum(oracle.forward_scaled(*ps[b], c) for c in chunks)
        assert rel_err(got[b], want) < TOL, k
        assert rel_err(forward_chunks([f.handle for f in fw], *ps[b]), want) < TOL


def test_text_file_constructor(oracle, hmm_params, example_pairs, tmp_path):
    """Forwarder(input_filename, NSYM) on the reference's own text format (hmm.py:12-14)."""
    obs = example_pairs["hg18__bonobo"][:20000]
    p = tmp_path / "pair.ziphmm"
    p.write_text(" ".join(str(int(s)) for s in obs) + " ")     # prepare-alignments.py:99-105 writes "%d "
    f = Forwarder(str(p), NSYM=3)
    assert len(f) == obs.size
    pi, T, E = hmm_params("iso10_t0")
    assert rel_err(f.forward(pi, T, E), oracle.forward_scaled(pi, T, E, obs)) < TOL
    p2 = tmp_path / "nl.txt"
    p2.write_text("\n".join(str(int(s)) for s in obs[:100]))   # newline separated, no trailing space
    assert len(Forwarder(str(p2), 3)) == 100
    from imcoalhmm_amd import prepare
    p3 = tmp_path / "pair.imc"
    prepare.write_cache(str(p3), obs, 3)                       # packed 2-bit cache, same constructor
    g = Forwarder(str(p3), NSYM=3)
    assert len(g) == obs.size and g.forward(pi, T, E) == f.forward(pi, T, E)


@pytest.mark.parametrize("n", [1, 3, 4, 8, 10, 12, 16, 20, 23, 24, 28, 32, 37, 40, 48, 49, 64, 70, 150, 220, 256])
@pytest.mark.parametrize("mode", [2, 3], ids=["token-vector", "token-blocked"])
def test_compressed_path_all_kernel_shapes(oracle, n, mode):
    """Both token kernels for every shape whose operator table fits LDS (mode 3 falls back to the vector
    variant above N=24), mixed with short chunks that stay on the per-column kernel in the same call,
    with stitching forced (48-token segments)."""
    set_zip(mode)
    pi, T, E = synth.random_hmm(n, 3, seed=300 + n, stay=0.97)
    chunks = [compressible(L, seed=n * 10 + k) for k, L in enumerate((40_000, 5000, 4096, 100, 33_000))]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    assert fw[0].compressed_length()[0] < 40_000 / 3
    for seg in (48, 0):
        try:
            set_seg(seg)
            got = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
            plan = _capi.last_plan()
        finally:
            set_seg(0)
        assert plan["vector_tokens"] > 0 or n > 40                            # the token path ran (N>40: no table fits LDS)
        assert n > 64 or (mode == 3 and n > 24) or plan["vector_columns"] > 0 # ... and so did the raw-stream group
        if mode == 3 and n > 24:
            assert "k_big_propagate" in plan["kernels"]                       # GEMM-chain kernels for 24 < N <= 64
        for c, g in zip(chunks, got):
            want = oracle.forward_scaled(pi, T, E, c)
            assert rel_err(g, want) < TOL, (n, seg, c.size, g, want)
    set_zip(1)


@pytest.mark.parametrize("n", [25, 32, 33, 48, 57, 64])
def test_mid_size_gemm_chain_on_raw_stream(oracle, n):
    """24 < N <= 64 with the MFMA GEMM-chain kernels pinned (mode 5), raw symbol stream, forced stitching."""
    set_zip(5)
    pi, T, E = synth.random_hmm(n, 3, seed=900 + n, stay=0.97)
    chunks = [compressible(L, seed=n * 7 + k) for k, L in enumerate((3000, 17, 700))]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    try:
        set_seg(64)
        got = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
        assert "k_big_propagate" in _capi.last_plan()["kernels"]
    finally:
        set_seg(0)
        set_zip(1)
    for c, g in zip(chunks, got):
        assert rel_err(g, oracle.forward_scaled(pi, T, E, c)) < TOL, (n, c.size)


def test_compression_off_uses_percolumn_kernel(oracle, hmm_params):
    pi, T, E = hmm_params("iso20_t0")
    obs = synth.sample_alignment(pi, T, E, 100_000, seed=5)
    try:
        set_zip(0)
        f = Forwarder.from_array(obs, 3)
        a = f.forward(pi, T, E)
        assert _capi.last_plan()["vector_tokens"] == 0
        set_zip(1)
        g = Forwarder.from_array(obs, 3)
        b = g.forward(pi, T, E)
        plan = _capi.last_plan()
        assert plan["vector_tokens"] > 0 and plan["vector_columns"] == 0 and 3 < plan["token_alphabet"] <= 32
        assert f.forward(pi, T, E) == a                   # a chunk created uncompressed stays per-column
    finally:
        set_zip(1)
    want = oracle.forward_scaled(pi, T, E, obs)
    assert rel_err(a, want) < TOL and rel_err(b, want) < TOL


def test_impossible_sequence_and_nan(hmm_params):
    pi = np.array([0.5, 0.5]); T = np.array([[0.9, 0.1], [0.1, 0.9]])
    E = np.array([[1.0, 0.0], [1.0, 0.0]])
    obs = np.zeros(500, dtype=np.uint8); obs[250] = 1
    try:
        for seg in (0, 64):
            set_seg(seg)
            assert Forwarder.from_array(obs, 2).forward(pi, T, E) == -math.inf
            Tn = T.copy(); Tn[0, 0] = np.nan
            assert math.isnan(Forwarder.from_array(obs, 2).forward(pi, Tn, E))
    finally:
        set_seg(0)


def test_underflow_guard_long_rare_symbol_runs(oracle, zipmode):
    """Long runs of a very unlikely symbol: the power-of-two rescale must keep everything finite."""
    n = 20
    pi, T, E = synth.random_hmm(n, 3, seed=77, stay=0.9995)
    E[:, 1] = 1e-12
    E[:, 0] = 1.0 - 1e-12 - E[:, 2]
    obs = np.zeros(40_000, dtype=np.uint8)
    obs[1000:9000] = 1
    got = Forwarder.from_array(obs, 3).forward(pi, T, E)
    want = oracle.forward_scaled(pi, T, E, obs)
    assert math.isfinite(got) and rel_err(got, want) < TOL


def test_closed_form_at_full_size(hmm_params):
    """Size-independent property at BASELINE's 1e8-column scale is in test_gpu_fullsize.py;
    here a 4e6-column closed form (rank-one T) to keep this module fast."""
    n, nsym = 20, 3
    rng = np.random.default_rng(3)
    q = rng.random(n); q /= q.sum()
    T = np.tile(q, (n, 1))
    E = rng.random((n, nsym)); E /= E.sum(axis=1, keepdims=True)
    pi = rng.random(n); pi /= pi.sum()
    obs = rng.integers(0, nsym, size=4_000_000).astype(np.uint8)
    per_sym = np.log(q @ E)
    cnt = np.bincount(obs[1:], minlength=nsym)
    want = math.log(pi @ E[:, obs[0]]) + float(cnt @ per_sym)
    got = Forwarder.from_array(obs, nsym).forward(pi, T, E)
    assert rel_err(got, want) < 1e-10


def test_bad_arguments_raise(hmm_params):
    pi, T, E = hmm_params("iso10_t0")
    f = Forwarder.from_array(np.zeros(100, dtype=np.uint8), 3)
    with pytest.raises(ValueError):
        f.forward(pi, T[:5], E)
    with pytest.raises(ValueError):
        f.forward(pi, T, E[:, :2])           # chunk alphabet (3) larger than S=2
    with pytest.raises(ValueError):
        Forwarder.from_array(np.array([0, 5], dtype=np.uint8), 3)
    big = synth.random_hmm(257, 3, 1)
    with pytest.raises(ValueError):
        f.forward(*big)                      # N beyond the largest built kernel is refused loudly


def test_config1_end_to_end_from_theta(hmm_params_file, example_pairs, golden_loglik):
    """theta -> imcoalhmm_amd.models -> Likelihood -> device, against the golden cells whose (pi, T, E)
    came from the reference's own model classes (scripts/isolation-model.py:82-100 call pattern)."""
    from imcoalhmm_amd import models
    set_zip(1)
    fw = {k: Forwarder.from_array(v, 3) for k, v in example_pairs.items()}
    cases = [("iso10_t0", models.IsolationModel(10)), ("iso20_t0", models.IsolationModel(20)),
             ("im20_t0", models.IsolationMigrationModel(10, 10)), ("im150_t0", models.IsolationMigrationModel(75, 75))]
    for mkey, model in cases:
        theta = hmm_params_file[mkey + "_theta"]
        for pname in ("hg18__pantro2", "bonobo__ponabe2"):
            want = golden_loglik["%s|%s" % (pname, mkey)]["loglik"]
            got = Likelihood(model, fw[pname])(theta)
            assert rel_err(got, want) < 1e-10, (mkey, pname, got, want)
    # all six pairs as one likelihood, a population of proposals, invalid points gated
    model = models.IsolationModel(10)
    ll = Likelihood(model, list(fw.values()))
    theta = hmm_params_file["iso10_t0_theta"]
    total = sum(golden_loglik["%s|iso10_t0" % p]["loglik"] for p in fw)
    assert rel_err(ll(theta), total) < 1e-10
    thetas = np.stack([theta, theta * 1.1, -theta, theta * 0.9])
    vals = ll.batch(thetas)
    assert rel_err(vals[0], total) < 1e-10 and vals[2] == -np.inf
    assert rel_err(vals[1], ll(thetas[1])) < 1e-12 and rel_err(vals[3], ll(thetas[3])) < 1e-12


def test_maximum_likelihood_estimate_improves(hmm_params_file, example_pairs):
    """likelihood.py:36-87 driver: a short Nelder-Mead run from the script defaults must not end lower."""
    import io
    from imcoalhmm_amd import maximum_likelihood_estimate, models
    set_zip(1)
    ll = Likelihood(models.IsolationModel(10), Forwarder.from_array(example_pairs["hg18__pantro2"], 3))
    theta0 = hmm_params_file["iso10_t0_theta"]
    calls = []

    def counted(theta):
        calls.append(1)
        return ll(np.asarray(theta)) if len(calls) < 60 else -1e300      # bound the run

    log = io.StringIO()
    best = maximum_likelihood_estimate(counted, theta0, log_file=log)
    assert best.shape == theta0.shape and ll(best) >= ll(theta0)
    assert len(log.getvalue().splitlines()[0].split("\t")) == 3


@pytest.mark.parametrize("n", [70, 100, 140, 150, 192, 210, 256])
@pytest.mark.parametrize("mode", [2, 4], ids=["tokens", "raw"])
def test_matvec_chain_kernel_large_n(oracle, n, mode):
    """N > 64 with one segment per chunk (pinned by the 'vector' modes): the mat-vec chain kernel, batch of 3."""
    set_zip(mode)
    rng = np.random.default_rng(n)
    hmms = [synth.random_hmm(n, 3, seed=40 * n + b, stay=0.9 + 0.03 * b) for b in range(3)]
    pis, Ts, Es = (np.stack([h[k] for h in hmms]) for k in range(3))
    lens = (33_000, 1, 2, 17, 5000, 40_000, 16, 4097, 900, 33, 2500)
    chunks = [compressible(L, seed=n * 13 + k) for k, L in enumerate(lens)]
    chunks[4][1000:1400] = 2                       # a long run of the rare symbol
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    try:
        got = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
        plan = _capi.last_plan()
    finally:
        set_zip(1)
    assert "k_big_vector" in plan["kernels"] and "k_big_propagate" not in plan["kernels"]
    for b in range(3):
        for k in rng.choice(len(chunks), size=5, replace=False) if n > 100 else range(len(chunks)):
            want = oracle.forward_scaled(pis[b], Ts[b], Es[b], chunks[k])
            assert rel_err(got[b][k], want) < TOL, (n, mode, b, k, got[b][k], want)


@pytest.mark.parametrize("n", [32, 41, 64])
def test_matvec_chain_kernel_mid_n_many_chunks(oracle, n):
    """24 < N <= 64: with many long chunks x proposals the planner drops the transfer operators altogether.  (640 chains:
    with round 3's measured GEMM-chain step costs - profiles/r03_e_calib_big.txt - the 80 chains this test had until
    then are rightly put on the GEMM chain, which is 3-9x faster there.)"""
    set_zip(1)
    hmms = [synth.random_hmm(n, 3, seed=77 * n + b, stay=0.95) for b in range(16)]
    pis, Ts, Es = (np.stack([h[k] for h in hmms]) for k in range(3))
    chunks = [compressible(210_000 + 1000 * k, seed=n * 17 + k) for k in range(40)]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    got = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
    plan = _capi.last_plan()
    assert "k_big_vector" in plan["kernels"], plan["kernels"]
    try:
        set_zip(3, reset=False)                    # same data through the GEMM-chain kernels
        ref = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
        assert "k_big_propagate" in _capi.last_plan()["kernels"]
    finally:
        set_zip(1)
    assert np.max(np.abs(got / ref - 1)) < TOL
    for b, k in ((0, 0), (1, 39), (15, 20)):
        assert rel_err(got[b][k], oracle.forward_scaled(pis[b], Ts[b], Es[b], chunks[k])) < TOL


@pytest.mark.parametrize("n", [28, 70])
def test_wide_token_levels(oracle, n):
    """Dictionaries beyond 256 tokens (16-bit streams) on the global-memory-table kernels: a 3e6-column first
    chunk trains ~3500 tokens; GEMM chain (automatic and forced segmentation) and mat-vec chain kernels."""
    set_zip(1)
    pi, T, E = synth.random_hmm(n, 3, seed=5000 + n, stay=0.97)
    chunks = [compressible(3_000_000, seed=5), compressible(50_000, seed=6), compressible(7, seed=7),
              compressible(300_000, seed=8)]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    ntok, alpha = fw[0].compressed_length(16384)
    assert alpha > 1024 and ntok < 3_000_000 / 14
    assert fw[1].compressed_length(16384)[1] == alpha           # later chunks share the dictionary
    results = {}
    # N <= 64: mode 3 pins the GEMM-chain kernels (a 7-column chunk keeps the automatic choice off them)
    runs = (("auto", 1, 0), ("segmented", 1, 4096), ("matvec", 2, 0)) if n > 64 else (("gemm", 3, 0), ("segmented", 3, 4096))
    for label, mode, seg in runs:
        try:
            set_zip(mode, reset=False)
            set_seg(seg)
            results[label] = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
            plan = _capi.last_plan()
        finally:
            set_seg(0)
            set_zip(1, reset=False)
        assert "k_big" in plan["kernels"], plan["kernels"]
        assert plan["token_alphabet"] > 256, plan
    for k in ((1, 2, 3) if n > 64 else (0, 1, 2, 3)):
        want = oracle.forward_scaled(pi, T, E, chunks[k])
        for label, got in results.items():
            assert rel_err(got[k], want) < TOL, (n, label, k, got[k], want)
    try:                                                        # the long chunk against the raw symbol stream
        set_zip(0, reset=False)
        raw = fw[0].forward(pi, T, E)
    finally:
        set_zip(1)
    for label, got in results.items():
        assert rel_err(got[0], raw) < TOL, (n, label)


@pytest.mark.parametrize("n,mode", [(3, 1), (10, 0), (20, 1), (20, 2), (28, 3), (40, 1), (70, 1), (70, 5), (150, 1)])
def test_state_export_split_alignment(oracle, n, mode):
    """imc_forward_state: one alignment cut into contiguous slices (as `dist.SplitAlignmentLikelihood` does
    across GPUs) - vector from the first slice, exact transfer operators from the others - recombines to the
    log-likelihood of the whole alignment."""
    from imcoalhmm_amd.hmm import combine_states, forward_states
    set_zip(mode)
    hmms = [synth.random_hmm(n, 3, seed=8100 + 3 * n + b, stay=0.95) for b in range(2)]
    pis, Ts, Es = (np.stack([h[k] for h in hmms]) for k in range(3))
    whole = compressible(90_000 if n <= 70 else 30_000, seed=n + 1)
    cuts = [0, 40_000, 40_001, 41_000, whole.size] if n <= 70 else [0, 12_000, 12_001, whole.size]
    pieces = [whole[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    fw = [Forwarder.from_array(p, 3) for p in pieces]
    fw_whole = Forwarder.from_array(whole, 3)
    for seg in (0, 256) + ((4096,) if mode == 5 else ()):     # (mode 5 + 4096: GEMM chain with the rank-one hand-off)
        try:
            set_seg(seg)
            vec, vexp = forward_states([fw[0].handle], pis, Ts, Es, as_operator=False)
            ops, oexp = forward_states([f.handle for f in fw[1:]], pis, Ts, Es, as_operator=True)
            want_gpu = forward_chunks_batch([fw_whole.handle], pis, Ts, Es)
        finally:
            set_seg(0)
        assert vec.shape == (2, 1, n) and ops.shape == (2, len(fw) - 1, n, n) and oexp.shape == (2, len(fw) - 1, n)
        for b in range(2):
            got = combine_states(vec[b, 0], vexp[b, 0], ops[b], oexp[b])
            want = oracle.forward_scaled(pis[b], Ts[b], Es[b], whole)
            assert rel_err(got, want) < TOL, (n, mode, seg, b, got, want)
            assert rel_err(want_gpu[b], want) < TOL
            # the exported vector alone reproduces the first slice's own log-likelihood
            first = combine_states(vec[b, 0], vexp[b, 0], [], [])
            assert rel_err(first, oracle.forward_scaled(pis[b], Ts[b], Es[b], pieces[0])) < TOL
    # a one-column operator is diag(E[:,o]) T'
    P, pe = forward_states([fw[1].handle], pis[:1], Ts[:1], Es[:1], as_operator=True)
    assert pieces[1].size == 1
    o = int(pieces[1][0])
    assert np.allclose(np.ldexp(P[0, 0], pe[0, 0][None, :]), Es[0][:, o][:, None] * Ts[0].T, rtol=1e-13, atol=0)
    empty = Forwarder.from_array(np.zeros(0, dtype=np.uint8), 3)
    with pytest.raises(ValueError):
        forward_states([empty.handle], pis, Ts, Es, True)
    set_zip(1)


def _simulated_collapse(pi, T, E, chunk, offset, head):
    """max_ic |P[i][c] s* / (P[i][c*] s_c) - 1| of the exact transfer operator of chunk[offset:offset+head], in numpy
    fp64: what k_rank1_check evaluates on the device (kernels_big.hpp), independent of any device arithmetic."""
    C = [E[:, s][:, None] * T.T for s in range(E.shape[1])]
    P = np.eye(len(pi))
    for t in range(head):
        P = C[int(chunk[offset + t])] @ P
        if t % 16 == 15:
            P /= P.max()
    s = P.sum(axis=0)
    cs = int(np.argmax(s))
    with np.errstate(divide="ignore", invalid="ignore"):
        r = (P * s[cs]) / (P[:, [cs]] * s[None, :])
    return float(np.nanmax(np.abs(r - 1.0)))


@pytest.mark.parametrize("n,stay", [(28, 0.9), (70, 0.9), (150, 0.9), (150, 0.95), (70, 0.99999), (70, -1.0)])
def test_rank_one_handoff(oracle, n, stay):
    """GEMM chain with long segments: operators that provably collapsed to rank one (fast-mixing HMM) finish on the
    mat-vec chain, slow-mixing ones stay on the GEMM chain; either way the result is the oracle's.

    Whether a parameter set collapses within the 1024-column head is a property of the HMM, not of `stay` alone:
    the operators are diag(E[:,o]) T', whose contraction rate is roughly (second largest / largest) emission
    weight of the dominant symbol times the stay probability.  random_hmm(150, seed 9150) has its two largest
    E[:,0] within 1 % of each other and is still at |r-1| ~ 1e-3 after 1024 columns, while seed 9151 is at 1e-14
    (round 1 saw exactly this as "(76, 38)").  So the expectation is computed, per parameter set, by simulating the
    test of k_rank1_check in numpy on a few segments; parameter sets are then run one at a time and the device's
    count must be all-or-nothing accordingly, and the two-set batch must report the sum."""
    set_zip(1)
    if stay < 0:        # uninformative emissions + sticky transitions: the operator stays close to the identity
        hmms = [synth.random_hmm(n, 3, seed=9000 + n + b, stay=0.99999) for b in range(2)]
        hmms = [(pi, T, np.tile(np.array([0.9, 0.06, 0.04]), (n, 1))) for pi, T, _ in hmms]
    else:
        hmms = [synth.random_hmm(n, 3, seed=9000 + n + b, stay=stay) for b in range(2)]
    pis, Ts, Es = (np.stack([h[k] for h in hmms]) for k in range(3))
    chunks = [compressible(120_000, seed=n + 11), compressible(40_000, seed=n + 12)]
    seg, head = 4096, 1024                            # forced segment length; the head is a quarter of it
    expect = []
    for b in range(2):
        sims = [_simulated_collapse(pis[b], Ts[b], Es[b], chunks[0], k * seg, head) for k in (1, 7, 19)]
        expect.append("all" if max(sims) < 2.0 ** -46 else "none" if min(sims) > 2.0 ** -38 else "any")
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    handles = [f.handle for f in fw]
    try:
        set_zip(5, reset=False)                       # raw column stream on the GEMM chain: segments of 4096 columns
        set_seg(seg)
        got = forward_chunks_batch(handles, pis, Ts, Es, per_chunk=True)
        plan = _capi.last_plan()
        both = _capi.last_rank1()
        single = []
        for b in range(2):
            one = forward_chunks_batch(handles, pis[b:b + 1], Ts[b:b + 1], Es[b:b + 1], per_chunk=True)
            assert np.array_equal(one[0], got[b])     # the decision per (parameter set, segment) is deterministic
            single.append(_capi.last_rank1())
    finally:
        set_seg(0)
        set_zip(1)
    assert "rank1-handoff" in plan["kernels"], plan["kernels"]
    n_op_segs = sum(-(-len(c) // seg) - 1 for c in chunks)
    for b in range(2):
        checked, collapsed = single[b]
        assert checked == n_op_segs
        assert {"all": collapsed == checked, "any": True, "none": collapsed == 0}[expect[b]], (b, expect, single)
    assert both == (single[0][0] + single[1][0], single[0][1] + single[1][1]), (both, single)
    for b in range(2):
        for k, c in enumerate(chunks):
            want = oracle.forward_scaled(pis[b], Ts[b], Es[b], c)
            assert rel_err(got[b][k], want) < TOL, (n, stay, b, k, got[b][k], want)


def _skewed_symbols(n, nsym, seed):
    """A compressible stream over a large alphabet: three frequent symbols (one of them the last), the rest rare."""
    rng = np.random.default_rng(seed)
    hot = np.array([0, nsym // 2, nsym - 1])
    obs = np.where(rng.random(n) < 0.9, hot[rng.integers(0, 3, size=n)], rng.integers(0, nsym, size=n))
    return obs.astype(np.int32)


@pytest.mark.parametrize("nsym", [65, 257])
@pytest.mark.parametrize("n", [10, 20, 40, 150])
def test_ils_sized_alphabets(oracle, n, nsym):
    """the triplet (65) and quartet (257 symbols, scripts/prepare-alignments.py:142-146,
    186-190) alphabets of the ILS model.  Symbol 256 needs 16-bit observations; chunks this long are pair-compressed
    whatever the alphabet (byte tokens for 65 symbols, 16-bit rounds for both), and the raw stream is evaluated too."""
    pi, T, E = synth.random_hmm(n, nsym, seed=1000 + n + nsym, stay=0.95)
    chunks = [_skewed_symbols(m, nsym, seed=n + nsym + k) for k, m in enumerate((60_000, 1, 4099, 33))]
    want = [oracle.forward_scaled(pi, T, E, c) for c in chunks]
    try:
        for mode in (1, 0):
            set_zip(mode)
            fw = [Forwarder.from_array(c, nsym) for c in chunks]
            if mode == 1:
                ntok, alpha = fw[0].compressed_length(16384)
                assert alpha > nsym and ntok * 2 < len(chunks[0]), (ntok, alpha)     # the long chunk did compress
            for seg in (0, 256):
                set_seg(seg)
                per = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
                for g, w in zip(per, want):
                    assert rel_err(g, w) < TOL, (n, nsym, mode, seg, g, w, _capi.last_plan()["kernels"])
            set_seg(0)
            if mode == 1 and n > 64:
                assert "tokens" in _capi.last_plan()["kernels"]                     # the global-table kernels (N > 64) run on the tokens
    finally:
        set_seg(0)
        set_zip(1)


def test_quartet_alphabet_text_file_and_errors(tmp_path, oracle):
    """The 257-symbol alphabet through the reference's constructor (text file, hmm.py:12-16) and the API's limits."""
    nsym, n = 257, 12
    pi, T, E = synth.random_hmm(n, nsym, seed=3, stay=0.9)
    obs = _skewed_symbols(5000, nsym, seed=9)
    path = tmp_path / "quartet.txt"
    path.write_text(" ".join(str(int(s)) for s in obs) + " ")
    got = Forwarder(str(path), NSYM=nsym).forward(pi, T, E)
    assert rel_err(got, oracle.forward_scaled(pi, T, E, obs)) < TOL
    with pytest.raises(ValueError):
        Forwarder.from_array(np.array([0, 257], dtype=np.int32), nsym)              # symbol out of range
    with pytest.raises(ValueError):
        Forwarder.from_array(np.zeros(4, dtype=np.uint8), 300)                        # bytes cannot carry 300 symbols
    with pytest.raises(ValueError):
        Forwarder.from_array(np.zeros(4, dtype=np.int32), 5000)                       # beyond the library's limit


def test_forwarder_attributes_of_the_reference(example_pairs):
    """hmm.py:15-16 keeps (new_obs, sym2pair, new_nsyms) on the Forwarder: expanding the compressed stream through
    the pair table must give back the original observations, and position 0 is never merged."""
    set_zip(1)
    obs = np.tile(example_pairs["hg18__pantro2"], 2)
    f = Forwarder.from_array(obs, 3)
    assert f.NSYM == 3 and f.new_nsyms > 3
    pairs, new_obs = f.sym2pair, f.new_obs
    assert sorted(pairs) == list(range(3, f.new_nsyms)) and len(new_obs) == f.compressed_length(1 << 30)[0] < len(obs) // 8
    assert new_obs[0] == obs[0]

    def expand(z):
        out, stack = [], [int(z)]
        while stack:
            t = stack.pop()
            if t < 3:
                out.append(t)
            else:
                stack.extend((pairs[t][1], pairs[t][0]))
        return out
    table = {z: expand(z) for z in range(f.new_nsyms)}
    back = np.fromiter((s for z in new_obs for s in table[int(z)]), dtype=np.uint8)
    assert np.array_equal(back, obs)
    short = Forwarder.from_array(obs[:100], 3)                     # too short to compress: the raw stream itself
    assert short.new_nsyms == 3 and short.sym2pair == {} and np.array_equal(short.new_obs, obs[:100])


@pytest.mark.parametrize("n", [4, 8, 10, 13, 16, 20])
@pytest.mark.parametrize("seg", [0, 52, 1000])
@pytest.mark.parametrize("stream", [0, 1])
def test_hybrid_table_kernel(oracle, n, seg, stream):
    """k_zpropagate4: a dictionary level far beyond what LDS holds (up to 4096 tokens), operators in a global table -
    stream = 0: the hottest cached in LDS, the others streamed a step ahead; stream = 1: every operator streamed.
    Long compressible chunks so that the dictionary grows past LDS; ragged / tiny / empty chunks ride along; two
    parameter sets; forced segment lengths that are not multiples of 16 (the masked first / last blocks) and
    4-token granularity."""
    L = _capi.lib()
    hmms = [synth.random_hmm(n, 3, seed=4000 + n + b, stay=0.995) for b in range(2)]
    pis, Ts, Es = (np.stack([h[k] for h in hmms]) for k in range(3))
    chunks = [synth.sample_alignment(pis[0], Ts[0], Es[0], m, seed=70 + k) for k, m in enumerate((1_500_000, 0, 1, 37, 4099, 300_000))]
    want = [[oracle.forward_scaled(pis[b], Ts[b], Es[b], c) for c in chunks] for b in range(2)]
    try:
        set_zip(3)                                        # pinned: register-blocked kernel, fresh dictionary
        _capi.check(L.imc_set_blocked_kernel(5))          # hybrid table wherever a level beyond LDS exists
        _capi.check(L.imc_set_table_streaming(stream))
        fw = [Forwarder.from_array(c, 3) for c in chunks]
        assert fw[0].compressed_length(256)[1] > 128      # the dictionary did grow (16-bit levels beyond 256 tokens too)
        set_seg(seg)
        got = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
        kernels = _capi.last_plan()["kernels"]
        if n > 8:                                         # (up to N = 8 every byte level fits LDS: plain k_zpropagate3)
            assert "k_zpropagate4" in kernels and ("streamed" in kernels) == bool(stream), kernels
        # ... and pinned to every dictionary level up to 1024 tokens (byte streams to 256, 16-bit ids beyond)
        seen = set()
        for lvl in (7, 9, 10, 11, 12, 14):
            os.environ["IMC_FORCE_LEVEL"] = str(lvl)
            try:
                forced = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
            finally:
                del os.environ["IMC_FORCE_LEVEL"]
            seen.add(_capi.last_plan()["kernels"])
            for b in range(2):
                for k in range(len(chunks)):
                    w = want[b][k]
                    assert (forced[b][k] == 0.0 and w == 0.0) or rel_err(forced[b][k], w) < TOL, (n, seg, lvl, b, k, _capi.last_plan())
        if n > 8:
            assert any(",16" in k for k in seen), seen     # a 16-bit level ran on the hybrid kernel
        again = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
        assert np.array_equal(got, again)                 # bit-identical repeats
        _capi.check(L.imc_set_blocked_kernel(3))          # the LDS-table kernel on the same chunks agrees
        lds = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
        assert "k_zpropagate3" in _capi.last_plan()["kernels"]
    finally:
        set_seg(0)
        _capi.check(L.imc_set_blocked_kernel(4))
        _capi.check(L.imc_set_table_streaming(-1))
        set_zip(1)
    for b in range(2):
        for k in range(len(chunks)):
            w = want[b][k]
            assert (got[b][k] == 0.0 and w == 0.0) or rel_err(got[b][k], w) < TOL, (n, seg, b, k, got[b][k], w)
            assert (lds[b][k] == 0.0 and w == 0.0) or rel_err(lds[b][k], w) < TOL


def test_many_chunks_times_many_proposals_plan(oracle, hmm_params):
    """A GA / PSO population or 64 MC3 chains over many alignment files: chunks x parameter sets alone need more rounds of
    workgroups than the planner's search used to cover (every chunk takes at least one workgroup of 32 segment rows per
    parameter set).  Round 2's search then ran into its iteration limit and cut each chunk into TWO segments of thousands
    of tokens - 2 of a workgroup's 32 rows in use, 7x slower per proposal (100 x 1e6 columns, 64 proposals).  The plan
    must cut a chunk into (a multiple of) 32 segments, and the values must be right."""
    pi, T, E = hmm_params("iso10_t0")
    hm = [hmm_params("iso10_t%d" % (b % 3)) for b in range(72)]
    pis, Ts, Es = (np.stack([h[k] for h in hm]) for k in range(3))
    chunks = [synth.sample_alignment(pi, T, E, 60_000 + 500 * k, seed=700 + k) for k in range(70)]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    per = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
    plan = _capi.last_plan()
    longest = max(f.compressed_length(plan["token_alphabet"] or 256)[0] for f in fw) if plan["vector_tokens"] else max(len(c) for c in chunks)
    seglen = plan["token_segment_len"] or plan["column_segment_len"]
    assert seglen <= 2 * (longest // 32 + 16), (seglen, longest, plan)           # ~ longest / 32, not longest / 2
    assert plan["segments"] >= 16 * len(chunks), plan
    for b in (0, 1, 2, 71):
        for k in (0, 33, 69):
            assert rel_err(per[b][k], oracle.forward_scaled(pis[b], Ts[b], Es[b], chunks[k])) < TOL, (b, k)
    assert np.array_equal(per[0], per[3]) and np.array_equal(per[1], per[70])       # same parameter set, same bits


def test_every_number_of_parameter_sets_reaches_every_workgroup(oracle, hmm_params):
    """k_zpropagate4's XCD-affine grid (BigArgs::n_phases, csrc/kernels_big.hpp): a one-dimensional grid cut into phases -
    eight parameter sets at a time with one set per XCD, then the remaining B % 8 = 4a + 2b + c sets on two, four and eight
    XCDs each.  Every B from 1 to 26 (every combination of phases, with and without the eight-at-a-time phase), ragged
    chunks whose workgroup counts are not multiples of the XCDs per set, global-table kernel forced: every (set, chunk)
    value against the same set evaluated alone (1e-12: dictionary level and segment length depend on B), three sets against
    the oracle."""
    L = _capi.lib()
    hm = [hmm_params("iso20_t%d" % (b % 3)) for b in range(3)] + [synth.random_hmm(20, 3, seed=4100 + b, stay=0.99) for b in range(23)]
    chunks = [synth.sample_alignment(*hm[0], m, seed=4200 + k) for k, m in enumerate((1_500_000, 4_500, 33_333, 7_000))]
    try:
        _capi.check(L.imc_set_compression(3)); _capi.check(L.imc_set_blocked_kernel(5)); _capi.check(L.imc_dictionary_reset())
        fw = [Forwarder.from_array(c, 3) for c in chunks]
        handles = [f.handle for f in fw]
        alone = np.array([forward_chunks_batch(handles, *(x[None] for x in h), per_chunk=True)[0] for h in hm])
        assert "k_zpropagate4" in _capi.last_plan()["kernels"], _capi.last_plan()["kernels"]
        for b in range(3):
            for k, c in enumerate(chunks):
                assert rel_err(alone[b][k], oracle.forward_scaled(hm[b][0], hm[b][1], hm[b][2], c)) < TOL, (b, k)
        for B in range(2, 27):
            pis, Ts, Es = (np.stack([h[k] for h in hm[:B]]) for k in range(3))
            per = forward_chunks_batch(handles, pis, Ts, Es, per_chunk=True)
            plan = _capi.last_plan()
            assert "k_zpropagate4" in plan["kernels"], plan["kernels"]
            bad = np.argwhere(np.abs(per - alone[:B]) > 1e-12 * np.abs(alone[:B]))
            assert bad.size == 0, (B, bad[:4])
        del fw
    finally:
        _capi.check(L.imc_set_compression(1)); _capi.check(L.imc_set_blocked_kernel(4)); _capi.check(L.imc_dictionary_reset())


@pytest.mark.parametrize("n", [10, 20])
def test_many_short_chunks_share_workgroups(oracle, hmm_params, n):
    """Packed blocks (Z2Block::first == 2, csrc/imcoal_fwd.hip make_units): chunks that are a single segment take a slot each
    of a shared workgroup instead of a workgroup each.  Hundreds of short chunks of every length around the 16-token block
    size - runs of them broken by long chunks (which are cut and folded as before) and by empty ones - on the LDS-table and
    the global-table kernel, two parameter sets: every chunk against the oracle."""
    L = _capi.lib()
    hm = [hmm_params("iso%d_t%d" % (n, b)) for b in range(2)]
    pis, Ts, Es = (np.stack([h[k] for h in hm]) for k in range(3))
    rng = np.random.default_rng(77 + n)
    big = synth.sample_alignment(*hm[0], 900_000, seed=31 + n)
    lens = [int(x) for x in rng.integers(1, 6000, size=230)]
    for pos, m in ((0, 400_000), (57, 0), (58, 0), (120, 250_000), (121, 1), (229, 90_000)):
        lens[pos] = m
    offs = [int(rng.integers(0, big.size - m)) if m else 0 for m in lens]
    chunks = [big[o:o + m] for o, m in zip(offs, lens)]
    want = np.array([[oracle.forward_scaled(pis[b], Ts[b], Es[b], c) if c.size else 0.0 for c in chunks] for b in range(2)])
    try:
        for mode, variant, seg in ((1, 4, 0), (3, 3, 0), (3, 5, 0), (3, 5, 6000), (1, 4, 64)):
            _capi.check(L.imc_set_compression(mode)); _capi.check(L.imc_set_blocked_kernel(variant)); _capi.check(L.imc_dictionary_reset())
            fw = [Forwarder.from_array(c, 3) for c in chunks]
            recompress([f for f in fw if len(f)])
            _capi.check(L.imc_set_segment_length(seg))
            for rep in range(2):
                per = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es, per_chunk=True)
                bad = np.argwhere(np.abs(per - want) > TOL * np.abs(want))
                assert bad.size == 0, (n, mode, variant, seg, rep, bad[:5], _capi.last_plan()["kernels"])
            _capi.check(L.imc_set_segment_length(0))
            del fw
    finally:
        _capi.check(L.imc_set_segment_length(0)); _capi.check(L.imc_set_compression(1)); _capi.check(L.imc_set_blocked_kernel(4))
        _capi.check(L.imc_dictionary_reset())
