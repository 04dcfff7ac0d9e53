"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/imcoal_fwd.h declares, errors are loud (no CPU fallback), and the host shim keeps the
reference's Likelihood semantics (src/IMCoalHMM/likelihood.py:8-33).  No compute call is made."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

import imcoalhmm_amd
from imcoalhmm_amd import _capi, build
from imcoalhmm_amd.likelihood import Likelihood

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _capi.lib()


def test_header_symbols_all_exported(lib):
    hdr = "".join(open(os.path.join(REPO, "include", f)).read() for f in sorted(os.listdir(os.path.join(REPO, "include"))) if f.endswith(".h"))
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(imc_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    bound = {name for name, _, _ in _capi.SIGNATURES}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert getattr(lib, name) is not None


def test_version_and_device_count(lib):
    assert b"gfx950" in lib.imc_version()
    assert lib.imc_device_count() >= 0


def test_argument_errors_need_no_device(lib):
    h = ctypes.c_void_p()
    bad = np.array([0, 1, 3], dtype=np.uint8)
    rc = lib.imc_obs_create(bad.ctypes.data_as(_capi._u8p), bad.size, 3, ctypes.byref(h))
    assert rc == _capi.IMC_ERR_SYMBOL and b"symbol 3" in lib.imc_last_error()
    with pytest.raises(ValueError):
        _capi.check(rc)
    assert lib.imc_obs_create(bad.ctypes.data_as(_capi._u8p), 3, 0, ctypes.byref(h)) == _capi.IMC_ERR_ARG
    assert lib.imc_obs_create_from_text(b"/nonexistent/file", 3, ctypes.byref(h)) == _capi.IMC_ERR_IO
    neg = np.array([0, -1], dtype=np.int32)
    assert lib.imc_obs_create_i32(neg.ctypes.data_as(_capi._i32p), 2, 3, ctypes.byref(h)) == _capi.IMC_ERR_SYMBOL
    assert lib.imc_obs_free(None) == _capi.IMC_OK
    assert lib.imc_obs_length(None) == 0


def test_text_parser_rejects_garbage(lib, tmp_path):
    p = tmp_path / "bad.txt"
    p.write_text("0 1 x 2")
    h = ctypes.c_void_p()
    assert lib.imc_obs_create_from_text(os.fsencode(str(p)), 3, ctypes.byref(h)) == _capi.IMC_ERR_IO
    p.write_text("0 1 7 2")
    assert lib.imc_obs_create_from_text(os.fsencode(str(p)), 3, ctypes.byref(h)) == _capi.IMC_ERR_SYMBOL


def test_no_cpu_fallback_without_device(lib):
    if lib.imc_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(_capi.ImcError) as ei:
        imcoalhmm_amd.Forwarder.from_array(np.zeros(8, dtype=np.uint8), 3)
    assert ei.value.code == _capi.IMC_ERR_NODEVICE


def test_product_never_imports_oracle():
    """The product package must not route through the CPU oracle."""
    pkg = os.path.join(REPO, "imcoalhmm_amd")
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, fn)).read()
                assert "oracle" not in txt.replace("test infrastructure", "").lower() or fn in ("_capi.py",), fn
    assert "import oracle" not in open(os.path.join(pkg, "_capi.py")).read()


class _FakeModel(object):
    def __init__(self):
        self.built = []

    def valid_parameters(self, parameters):
        assert isinstance(parameters, np.ndarray)      # model.py:40
        return all(parameters > 0)                     # model.py:41

    def build_hidden_markov_model(self, parameters):
        self.built.append(tuple(parameters))
        n = 3
        return np.full(n, 1.0 / n), np.full((n, n), 1.0 / n), np.full((n, 2), 0.5) * parameters[0]


class _FakeForwarder(object):
    def __init__(self, scale):
        self.scale = scale
        self.calls = 0

    def forward(self, pi, T, E):
        self.calls += 1
        return self.scale * float(E[0, 0])


def test_likelihood_gate_and_sum_semantics():
    m = _FakeModel()
    fs = [_FakeForwarder(1.0), _FakeForwarder(10.0), _FakeForwarder(100.0)]
    ll = Likelihood(m, fs)
    assert ll(np.array([2.0, 1.0])) == sum(f.scale * 1.0 for f in fs)        # likelihood.py:33
    assert ll(np.array([-1.0, 1.0])) == -float("inf")                          # likelihood.py:29-30
    assert all(f.calls == 1 for f in fs)                                        # invalid theta never reaches forwarders
    single = Likelihood(m, _FakeForwarder(3.0))                                 # likelihood.py:22-25
    assert len(single.forwarders) == 1 and single(np.array([2.0])) == 3.0
    out = ll.batch([np.array([2.0, 1.0]), np.array([0.0, 1.0]), np.array([4.0, 1.0])])
    assert out[0] == 111.0 and out[1] == -math.inf and out[2] == 222.0


def test_synth_generator_is_deterministic(hmm_params):
    from imcoalhmm_amd import synth
    pi, T, E = hmm_params("iso20_t0")
    a = synth.sample_alignment(pi, T, E, 50_000, seed=20240001)
    b = synth.sample_alignment(pi, T, E, 50_000, seed=20240001)
    assert a.dtype == np.uint8 and a.size == 50_000 and (a == b).all()
    frac = np.bincount(a, minlength=3) / a.size
    assert frac[0] > 0.85 and 0.01 < frac[2] < 0.10


def test_pairwise_encoder_matches_reference_rule(lib):
    """scripts/prepare-alignments.py:99-105: 2 if either base not in ACGT, 0 if equal, 1 otherwise."""
    from imcoalhmm_amd import prepare
    rng = np.random.default_rng(0)
    alphabet = np.array(list("ACGTacgtN-nRY"))
    s1 = "".join(rng.choice(alphabet, size=5000))
    s2 = "".join(rng.choice(alphabet, size=5000))
    got = prepare.encode_pairwise(s1, s2)
    clean = set("ACGT")
    want = [2 if (a.upper() not in clean or b.upper() not in clean) else (0 if a.upper() == b.upper() else 1)
            for a, b in zip(s1, s2)]
    assert got.tolist() == want
    with pytest.raises(ValueError):
        prepare.encode_pairwise("ACG", "AC")


def test_text_and_cache_files_round_trip(lib, tmp_path, example_pairs):
    """Both on-disk formats parse back to the same symbols; the text one is the reference's own format."""
    from imcoalhmm_amd import prepare
    obs = example_pairs["hg18__pantro2"][:30001]
    t, c = str(tmp_path / "pair.ziphmm"), str(tmp_path / "pair.imc")
    prepare.write_text(t, obs)
    prepare.write_cache(c, obs, 3)
    assert open(t).read(12) == " ".join(str(int(s)) for s in obs[:6]) + " "
    assert os.path.getsize(c) < os.path.getsize(t) / 7            # 2 bits vs 2 bytes per column
    assert (prepare.read_observations(t, 3) == obs).all()
    assert (prepare.read_observations(c, 3) == obs).all()
    big = np.arange(200, dtype=np.uint8)                            # > 4 symbols: byte cache
    prepare.write_cache(c, big, 200)
    assert (prepare.read_observations(c, 256) == big).all()
    with pytest.raises(ValueError):
        prepare.read_observations(c, 100)                           # alphabet larger than requested nsym


def test_prepare_cli_on_fasta_and_phylip(lib, tmp_path):
    from imcoalhmm_amd import prepare
    fa = tmp_path / "x.fa"
    fa.write_text(">a desc\nACGTNAC\nGT\n>b\nACCTAAC\nGA\n>c\nAAAAAAAAA\n")
    out = tmp_path / "ab.txt"
    assert prepare.main([str(fa), str(out), "--names", "a,b"]) == 0
    assert out.read_text() == "0 0 1 0 2 0 0 0 1 "
    ph = tmp_path / "x.phy"
    ph.write_text(" 2 9\na         ACGTNACGT\nb         ACCTAACGA\n")
    out2 = tmp_path / "ab.imc"
    assert prepare.main([str(ph), str(out2), "--in-format", "phylip", "--cache"]) == 0
    assert prepare.read_observations(str(out2), 3).tolist() == [0, 0, 1, 0, 2, 0, 0, 0, 1]
