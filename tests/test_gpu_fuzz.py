"""Seeded randomized sweep over the whole dispatch space: N, alphabet, chunk mixes (empty / tiny / long,
compressible or not), stream + kernel modes, forced segment lengths and batch sizes - every value checked
against the CPU oracle.  Catches dispatch/planning bugs that the per-kernel tests cannot."""
import os

import numpy as np
import pytest

from conftest import rel_err
from imcoalhmm_amd import Forwarder, _capi, synth
from imcoalhmm_amd.hmm import forward_chunks_batch

pytestmark = pytest.mark.gpu
N_CASES = int(os.environ.get("IMC_FUZZ_CASES", "150"))     # soak runs: IMC_FUZZ_CASES=1500


def _chunk(rng, nsym, L):
    if L == 0:
        return np.zeros(0, dtype=np.uint8)
    if rng.random() < 0.6:                       # compressible: long runs of symbol 0
        p = np.full(nsym, 0.08 / max(nsym - 1, 1)); p[0] = 0.92
        return rng.choice(nsym, size=L, p=p / p.sum()).astype(np.uint8)
    return rng.integers(0, nsym, size=L).astype(np.uint8)


@pytest.mark.parametrize("case", range(N_CASES))
def test_random_dispatch(oracle, case):
    rng = np.random.default_rng(1000 + case)
    L = _capi.lib()
    n = int(rng.choice([1, 2, 3, 5, 8, 10, 13, 16, 20, 22, 24, 27, 32, 40, 47, 64, 65, 90, 100, 128, 140, 150, 192, 200, 256]))
    nsym = int(rng.choice([2, 3, 3, 3, 4, 7]))
    mode = int(rng.integers(0, 6))
    seg = int(rng.choice([0, 0, 16, 48, 256, 1000]))
    B = int(rng.choice([1, 1, 2, 3]))
    heavy = n > 64
    lens = [int(x) for x in rng.choice([0, 1, 5, 16, 17, 100, 999, 4096, 5000, 20000 if heavy else 60000,
                                       9000 if heavy else 250000], size=int(rng.integers(1, 5)))]
    hmms = [synth.random_hmm(n, nsym, seed=case * 10 + b, stay=float(rng.choice([0.5, 0.9, 0.999]))) for b in range(B)]
    chunks = [_chunk(rng, nsym, x) for x in lens]
    variant = int(rng.choice([2, 3, 4, 5, 5]))           # form of the register-blocked kernel (N <= 24): VALU, MFMA + LDS table,
    stream = int(rng.choice([-1, 0, 1]))                 # k_zpropagate4's table: automatic, LDS-cached hot set, streamed
    if n <= 24 and rng.random() < 0.3:                   # larger batches: every phase of k_zpropagate4's XCD-affine grid
        B = int(rng.choice([4, 5, 8, 9, 12, 17]))        # (drawn last: the cases of earlier rounds keep their other draws)
        hmms = [synth.random_hmm(n, nsym, seed=case * 10 + b, stay=float(rng.choice([0.5, 0.9, 0.999]))) for b in range(B)]
    try:                                                 # automatic, hybrid table wherever possible
        _capi.check(L.imc_set_blocked_kernel(variant))
        _capi.check(L.imc_set_table_streaming(stream))
        _capi.check(L.imc_set_compression(mode))
        _capi.check(L.imc_dictionary_reset())
        _capi.check(L.imc_set_segment_length(seg))
        fw = [Forwarder.from_array(c, nsym) for c in chunks]
        got = forward_chunks_batch([f.handle for f in fw], np.stack([h[0] for h in hmms]), np.stack([h[1] for h in hmms]),
                                   np.stack([h[2] for h in hmms]), per_chunk=True)
        kernels = _capi.last_plan()["kernels"]
    finally:
        L.imc_set_blocked_kernel(4)
        L.imc_set_table_streaming(-1)
        L.imc_set_compression(1)
        L.imc_set_segment_length(0)
    for b in range(B):
        for f, c in enumerate(chunks):
            want = oracle.forward_scaled(*hmms[b], c)
            g = got[b, f]
            assert (g == 0.0 and want == 0.0) or rel_err(g, want) < 1e-11, (case, n, nsym, mode, variant, stream, seg, B, lens, kernels, b, f, g, want)


@pytest.mark.parametrize("case", range(max(40, N_CASES // 4)))
def test_random_split_state(oracle, case):
    """imc_forward_state under the same random dispatch: a random alignment cut at random points, vector from the
    first piece and operators from the rest, recombined and checked against the oracle on the whole alignment."""
    from imcoalhmm_amd.hmm import combine_states, forward_states
    rng = np.random.default_rng(7000 + case)
    L = _capi.lib()
    n = int(rng.choice([1, 2, 5, 10, 16, 20, 24, 27, 40, 64, 65, 128, 150, 200]))
    nsym = int(rng.choice([2, 3, 3, 4]))
    mode = int(rng.integers(0, 6))
    seg = int(rng.choice([0, 0, 16, 256]))
    B = int(rng.choice([1, 2]))
    total = int(rng.choice([50, 3000, 12000 if n > 64 else 80000]))
    whole = _chunk(rng, nsym, total)
    cuts = sorted(set([0, total] + [int(x) for x in rng.integers(1, total, size=int(rng.integers(1, 4)))]))
    pieces = [whole[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    hmms = [synth.random_hmm(n, nsym, seed=case * 10 + b, stay=float(rng.choice([0.5, 0.9, 0.999]))) for b in range(B)]
    pis, Ts, Es = (np.stack([h[k] for h in hmms]) for k in range(3))
    try:
        _capi.check(L.imc_set_compression(mode))
        _capi.check(L.imc_dictionary_reset())
        _capi.check(L.imc_set_segment_length(seg))
        fw = [Forwarder.from_array(p, nsym) for p in pieces]
        vec, vexp = forward_states([fw[0].handle], pis, Ts, Es, False)
        if len(fw) > 1:
            ops, oexp = forward_states([f.handle for f in fw[1:]], pis, Ts, Es, True)
        kernels = _capi.last_plan()["kernels"]
    finally:
        L.imc_set_compression(1)
        L.imc_set_segment_length(0)
    for b in range(B):
        got = combine_states(vec[b, 0], vexp[b, 0], ops[b] if len(fw) > 1 else [], oexp[b] if len(fw) > 1 else [])
        want = oracle.forward_scaled(*hmms[b], whole)
        assert rel_err(got, want) < 1e-11, (case, n, nsym, mode, seg, B, cuts, kernels, b, got, want)


@pytest.mark.parametrize("case", range(max(24, N_CASES // 8)))
def test_random_handoff(oracle, case):
    """Long GEMM-chain segments with the certified rank-one hand-off: random state counts, stickiness (so that some
    segments collapse and some do not), segment lengths and batch sizes, against the oracle."""
    rng = np.random.default_rng(9100 + case)
    L = _capi.lib()
    n = int(rng.choice([28, 48, 64, 70, 100, 140, 150]))
    B = int(rng.choice([1, 2]))
    seg = int(rng.choice([2048, 4096, 8192]))
    lens = [int(rng.choice([30_000, 70_000, 150_000 if n <= 100 else 90_000])) for _ in range(int(rng.integers(1, 3)))]
    hmms = [synth.random_hmm(n, 3, seed=case * 10 + b, stay=float(rng.choice([0.5, 0.9, 0.99, 0.999, 0.9999]))) for b in range(B)]
    if rng.random() < 0.25:                                   # uninformative emissions: no collapse at all
        hmms = [(pi, T, np.tile(np.array([0.85, 0.1, 0.05]), (n, 1))) for pi, T, _ in hmms]
    chunks = [_chunk(rng, 3, x) for x in lens]
    try:
        _capi.check(L.imc_set_compression(5))                 # raw stream, GEMM chain pinned
        _capi.check(L.imc_set_segment_length(seg))
        fw = [Forwarder.from_array(c, 3) for c in chunks]
        got = forward_chunks_batch([f.handle for f in fw], np.stack([h[0] for h in hmms]), np.stack([h[1] for h in hmms]),
                                   np.stack([h[2] for h in hmms]), per_chunk=True)
        kernels = _capi.last_plan()["kernels"]
        stats = _capi.last_rank1()
    finally:
        L.imc_set_compression(1)
        L.imc_set_segment_length(0)
    assert "rank1-handoff" in kernels and stats[0] > 0, (kernels, stats)
    for b in range(B):
        for f, c in enumerate(chunks):
            want = oracle.forward_scaled(*hmms[b], c)
            assert rel_err(got[b, f], want) < 1e-11, (case, n, seg, B, lens, stats, b, f, got[b, f], want)
