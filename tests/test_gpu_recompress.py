"""imc_obs_recompress: a data set that arrives as many chunks gets ONE pair dictionary trained on all of them."""
import numpy as np
import pytest

from conftest import rel_err
from imcoalhmm_amd import Forwarder, Likelihood, _capi, synth
from imcoalhmm_amd.hmm import forward_chunks_batch, recompress

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("key,n", [("iso20_t0", 20), ("im150_t0", 150)])
def test_joint_dictionary_matches_oracle_and_compresses_better(oracle, hmm_params, key, n):
    pi, T, E = hmm_params(key)
    L = _capi.lib()
    _capi.check(L.imc_dictionary_reset())
    lens = (60_000, 45_000, 5_000, 33, 0, 80_000, 70_000, 64_000)
    chunks = [synth.sample_alignment(pi, T, E, m, seed=300 + k) for k, m in enumerate(lens)]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    h = [f.handle for f in fw]
    pis, Ts, Es = pi[None], T[None], E[None]
    before = forward_chunks_batch(h, pis, Ts, Es, per_chunk=True)[0]
    tokens_before = sum(f.compressed_length(1 << 30)[0] for f in fw)
    alphabet_before = fw[0].new_nsyms
    recompress(fw)
    after = forward_chunks_batch(h, pis, Ts, Es, per_chunk=True)[0]
    again = forward_chunks_batch(h, pis, Ts, Es, per_chunk=True)[0]
    assert np.array_equal(after, again)
    tokens_after = sum(f.compressed_length(1 << 30)[0] for f in fw)
    assert fw[0].new_nsyms >= alphabet_before and tokens_after < tokens_before     # the larger sample found more pairs
    for k, c in enumerate(chunks):
        want = oracle.forward_scaled(pi, T, E, c)
        for got in (before[k], after[k]):
            assert (got == 0.0 and want == 0.0) or rel_err(got, want) < 1e-11, (k, got, want)
    # the reference-shaped views follow the new dictionary; the decoded stream is still the alignment
    pairs = fw[0].sym2pair
    def expand(t):
        return [t] if t < 3 else expand(pairs[t][0]) + expand(pairs[t][1])
    dec = [s for t in fw[0].new_obs for s in expand(int(t))]
    assert np.array_equal(np.array(dec, dtype=np.uint8), chunks[0])
    # a chunk created afterwards shares the joint dictionary, and Likelihood() recompresses its forwarders itself
    late = Forwarder.from_array(chunks[1], 3)
    assert late.new_nsyms == fw[0].new_nsyms

    class M(object):
        def valid_parameters(self, theta): return True
        def build_hidden_markov_model(self, theta): return pi, T, E
    ll = Likelihood(M(), fw)                       # same forwarders again: nothing to retrain (same dictionary object)
    assert fw[0].new_nsyms == late.new_nsyms
    ll = Likelihood(M(), fw + [late])
    want_total = sum(oracle.forward_scaled(pi, T, E, c) for c in chunks) + oracle.forward_scaled(pi, T, E, chunks[1])
    assert rel_err(ll(np.zeros(3)), want_total) < 1e-11
    _capi.check(L.imc_dictionary_reset())


def test_recompress_edge_cases(oracle, hmm_params):
    pi, T, E = hmm_params("iso10_t0")
    L = _capi.lib()
    _capi.check(L.imc_dictionary_reset())
    _capi.check(L.imc_obs_recompress(None, 0))                         # nothing to do
    short = [Forwarder.from_array(synth.sample_alignment(pi, T, E, m, seed=5 + m), 3) for m in (0, 7, 900)]
    recompress(short)                                                  # too short to compress: left alone
    assert all(f.new_nsyms == 3 for f in short)
    one = Forwarder.from_array(synth.sample_alignment(pi, T, E, 50_000, seed=9), 3)
    recompress([one])                                                  # a single Forwarder: no-op
    def skewed(k):                                                     # mostly symbol 0: compressible
        r = np.random.default_rng(k)
        return np.where(r.random(40_000) < 0.9, 0, r.integers(0, 300, size=40_000)).astype(np.int32)
    wide = [Forwarder.from_array(skewed(k), 300) for k in range(3)]
    recompress(wide + short)                                           # 16-bit raw alphabet next to a byte alphabet
    assert wide[0].new_nsyms > 300
    hmm300 = synth.random_hmm(12, 300, seed=4)
    for k, f in enumerate(wide):
        want = oracle.forward_scaled(*hmm300, skewed(k))
        assert rel_err(f.forward(*hmm300), want) < 1e-11
    _capi.check(L.imc_dictionary_reset())


def test_joint_dictionary_does_not_depend_on_where_the_chunks_live(hmm_params):
    """The training sample of imc_obs_recompress is the concatenation of the chunks' heads.  Until round 3 the chunks were
    put in ADDRESS order first, so the dictionary - its size, the token counts, once even whether the byte phase filled its
    256 entries and the 16-bit levels existed at all (a bench run came out 25-40 % slow) - changed from run to run of the same
    program.  Same chunks, created twice with other allocations in between and handed over in another order: the same
    dictionary, pair for pair, and the same values bit for bit."""
    pi, T, E = hmm_params("iso10_t0")
    L = _capi.lib()
    chunks = [synth.sample_alignment(pi, T, E, 150_000 + 7_000 * k, seed=900 + k) for k in range(24)]
    results = []
    junk = []
    for attempt in range(3):
        _capi.check(L.imc_dictionary_reset())
        junk.append([Forwarder.from_array(chunks[k][:5000 + 999 * attempt], 3) for k in range(attempt * 3)])   # shifts the heap
        fw = [Forwarder.from_array(c, 3) for c in chunks]
        order = list(range(len(fw)))
        if attempt == 1:
            order = order[::-1]
        elif attempt == 2:
            order = order[5:] + order[:5]
        recompress([fw[k] for k in order])
        s2p = fw[0].sym2pair                                           # {token: (left, right)}
        pairs = sorted((int(t), tuple(int(x) for x in s2p[t])) for t in s2p)
        value = forward_chunks_batch([f.handle for f in fw], pi[None], T[None], E[None], per_chunk=True)[0]
        results.append((fw[0].new_nsyms, pairs, [f.compressed_length(1 << 30) for f in fw], value))
        del fw
    for other in results[1:]:
        assert other[0] == results[0][0] and other[1] == results[0][1] and other[2] == results[0][2]
        assert np.array_equal(other[3], results[0][3])
