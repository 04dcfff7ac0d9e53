"""world_size-2 gloo test of the multi-GPU sharding logic on CPU.

The per-rank partial sums are produced by the CPU oracle here (tests may use it as a stand-in for
the device); what is under test is the static round-robin partition and the all-reduce(sum)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imcoalhmm_amd import synth
    from imcoalhmm_amd.dist import DistributedLikelihood, shard_indices
    from oracle import oracle_lib
    d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
    pi, T, E = d["iso10_t0_pi"], d["iso10_t0_T"], d["iso10_t0_E"]
    pi2, T2, E2 = d["iso10_t1_pi"], d["iso10_t1_T"], d["iso10_t1_E"]
    n_chunks = 7
    chunks = [synth.sample_alignment(pi, T, E, 2000 + 300 * k, seed=20240100 + k) for k in range(n_chunks)]
    mine = shard_indices(n_chunks, rank, world)

    def local_eval(pis, Ts, Es):
        vals = [sum(oracle_lib.forward_scaled(pis[b], Ts[b], Es[b], chunks[i]) for i in mine)
                for b in range(pis.shape[0])]
        return torch.tensor(vals, dtype=torch.float64)

    class M(object):
        def valid_parameters(self, p):
            return all(p > 0)

        def build_hidden_markov_model(self, p):
            return (pi, T, E) if p[0] < 2 else (pi2, T2, E2)

    ll = DistributedLikelihood(M(), [object() for _ in mine], local_eval=local_eval)
    single = ll(np.array([1.0]))
    batch = ll.batch([np.array([1.0]), np.array([-1.0]), np.array([3.0])])
    want1 = sum(oracle_lib.forward_scaled(pi, T, E, c) for c in chunks)
    want2 = sum(oracle_lib.forward_scaled(pi2, T2, E2, c) for c in chunks)
    q.put((rank, mine, single, batch.tolist(), want1, want2))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_likelihood_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    shards = sorted(sum((r[1] for r in res), []))
    assert shards == list(range(7))                       # a partition: every chunk exactly once
    for rank, mine, single, batch, want1, want2 in res:
        assert abs(single - want1) / abs(want1) < 1e-13    # every rank sees the global value
        assert abs(batch[0] - want1) / abs(want1) < 1e-13
        assert batch[1] == -float("inf")                   # invalid theta gate (likelihood.py:29-30)
        assert abs(batch[2] - want2) / abs(want2) < 1e-13
    assert res[0][2] == res[1][2]                          # identical on all ranks


def test_shard_indices_cover_and_balance():
    from imcoalhmm_amd.dist import shard_indices
    for n in (0, 1, 7, 256):
        for w in (1, 2, 4, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _numpy_state(pi, T, E, obs, as_operator):
    """Reference slice state with plain numpy: scaled vector from pi, or scaled transfer operator."""
    n = pi.shape[0]
    if as_operator:
        P = np.eye(n)
        exps = np.zeros(n, dtype=np.int64)
        for o in obs:
            P = (E[:, o][:, None] * T.T) @ P
            m = P.max(axis=0)
            sh = np.frexp(m)[1]
            P = np.ldexp(P, -sh[None, :])
            exps += sh
        return P, exps
    a = pi * E[:, obs[0]]
    e = 0
    for o in obs[1:]:
        a = E[:, o] * (T.T @ a)
        sh = int(np.frexp(a.max())[1])
        a = np.ldexp(a, -sh)
        e += sh
    return a, e


def _split_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imcoalhmm_amd import synth
    from imcoalhmm_amd.dist import SplitAlignmentLikelihood, slice_bounds
    from oracle import oracle_lib
    d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
    hm = [(d["iso10_t%d_pi" % k], d["iso10_t%d_T" % k], d["iso10_t%d_E" % k]) for k in (0, 1)]
    whole = synth.sample_alignment(*hm[0], 6001, seed=77)
    lo, hi = slice_bounds(whole.size, rank, world)
    mine = whole[lo:hi]

    def local_state(pis, Ts, Es, as_operator):
        vals, exps = zip(*[_numpy_state(pis[b], Ts[b], Es[b], mine, as_operator) for b in range(pis.shape[0])])
        return np.stack(vals), np.stack([np.asarray(x) for x in exps])

    class M(object):
        def valid_parameters(self, p):
            return all(p > 0)

        def build_hidden_markov_model(self, p):
            return hm[0] if p[0] < 2 else hm[1]

    ll = SplitAlignmentLikelihood(M(), object(), local_state=local_state)
    got = [ll(np.array([1.0])), ll(np.array([3.0])), ll(np.array([-1.0]))]
    pis, Ts, Es = (np.stack([h[k] for h in hm]) for k in range(3))
    batch = ll.forward_params_batch(pis, Ts, Es)
    want = [oracle_lib.forward_scaled(*h, whole) for h in hm]
    q.put((rank, (lo, hi), got, batch.tolist(), want))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_alignment_world(world):
    """One alignment cut into contiguous slices, one per rank: vector from rank 0, operators from the others,
    all_gather + ordered combine (dist.SplitAlignmentLikelihood) equals the single-chain forward."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bounds = sorted(r[1] for r in res)
    assert bounds[0][0] == 0 and bounds[-1][1] == 6001
    assert all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:]))       # contiguous, no overlap
    for rank, _, got, batch, want in res:
        assert abs(got[0] - want[0]) / abs(want[0]) < 1e-12
        assert abs(got[1] - want[1]) / abs(want[1]) < 1e-12
        assert got[2] == -float("inf")
        assert abs(batch[0] - want[0]) / abs(want[0]) < 1e-12 and abs(batch[1] - want[1]) / abs(want[1]) < 1e-12
    assert all(r[2][:2] == res[0][2][:2] for r in res)                     # identical on all ranks


def _proposal_worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imcoalhmm_amd import models, synth
    from imcoalhmm_amd.dist import ProposalShardedLikelihood
    from oracle import oracle_lib
    model = models.IsolationModel(6)
    theta0 = np.array([0.001, 1000.0, 0.4])
    chunks = [synth.sample_alignment(*model.build_hidden_markov_model(theta0), 1500 + 200 * k, seed=900 + k) for k in range(2)]
    calls = []

    def local_eval(pis, Ts, Es):
        calls.append(pis.shape[0])
        return np.array([sum(oracle_lib.forward_scaled(pis[b], Ts[b], Es[b], c) for c in chunks) for b in range(pis.shape[0])])

    ll = ProposalShardedLikelihood(model, [object(), object()], local_eval=local_eval)
    thetas = [theta0 * (1 + 0.05 * k) for k in range(5)] + [-theta0]
    got = ll.batch(thetas)
    want = [sum(oracle_lib.forward_scaled(*model.build_hidden_markov_model(t), c) for c in chunks) for t in thetas[:5]]
    q.put((rank, got.tolist(), want, calls, ll(theta0)))
    dist.barrier()
    dist.destroy_process_group()


def test_proposal_sharding_world2():
    """Few chunks, many proposals: ranks split the parameter sets (second axis) and all_gather the values."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_proposal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, want, calls, single in res:
        assert all(abs(g - w) / abs(w) < 1e-13 for g, w in zip(got[:5], want))
        assert got[5] == -float("inf")
        assert abs(single - want[0]) / abs(want[0]) < 1e-13
    assert sorted(r[3][0] for r in res) == [2, 3]          # 5 valid proposals split 3 + 2
    assert res[0][1] == res[1][1]


def _worker_ordered(rank, world, port, q):
    """Ragged chunk lengths, shards balanced by column count, per-rank partials combined in fixed rank order."""
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imcoalhmm_amd.dist import DistributedLikelihood, shard_indices
    lengths = [5000, 100, 2500, 2500, 40, 1200, 3000, 700, 60, 1800, 2200]
    mine = shard_indices(len(lengths), rank, world, lengths=lengths)
    # partial sums whose floating-point total depends on the order of the additions
    vals = np.array([1e16, 1.0, -1e16, 3.0, 1e-3, 7.0, 2.0 ** -40, 5.0, -2.0, 1e8, 1.0 / 3.0])

    def local_eval(pis, Ts, Es):
        out = []
        for b in range(pis.shape[0]):
            t = 0.0
            for i in mine:
                t += float(vals[i]) * (b + 1)
            out.append(t)
        return torch.tensor(out, dtype=torch.float64)

    class M(object):
        def valid_parameters(self, p):
            return True

        def build_hidden_markov_model(self, p):
            return np.ones(2) / 2, np.eye(2), np.ones((2, 3)) / 3

    pis, Ts, Es = np.ones((3, 2)) / 2, np.stack([np.eye(2)] * 3), np.ones((3, 2, 3)) / 3
    ordered = DistributedLikelihood(M(), [], local_eval=local_eval, reduction="ordered").forward_params_batch(pis, Ts, Es)
    plain = DistributedLikelihood(M(), [], local_eval=local_eval).forward_params_batch(pis, Ts, Es)
    q.put((rank, mine, ordered.tolist(), plain.tolist(), local_eval(pis, Ts, Es).tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ordered_reduction_and_balanced_shards(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ordered, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lengths = [5000, 100, 2500, 2500, 40, 1200, 3000, 700, 60, 1800, 2200]
    shards = [r[1] for r in res]
    assert sorted(sum(shards, [])) == list(range(len(lengths)))          # a partition
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert max(loads) <= 4 * (sum(lengths) / world) / 3 + 1                # LPT bound; round-robin would not meet it
    assert max(loads) - min(loads) <= max(lengths)
    # every rank holds the same bits, and they are the rank-ordered host sum of the partials
    for b in range(3):
        want = 0.0
        for r in res:
            want += r[4][b]
        assert all(r[2][b] == want for r in res), (b, [r[2][b] for r in res], want)
    # (the plain all-reduce agrees to rounding, the ordered one exactly)
    for r in res:
        assert np.allclose(r[3], r[2], rtol=1e-12, atol=2.0)


def test_shard_indices_properties():
    sys.path.insert(0, REPO)
    from imcoalhmm_amd.dist import shard_indices
    rng = np.random.default_rng(4)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 64):
            assert sorted(sum((shard_indices(n, r, world) for r in range(world)), [])) == list(range(n))
            lengths = rng.integers(1, 10_000_000, size=n).tolist()
            parts = [shard_indices(n, r, world, lengths=lengths) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            if n >= world:
                loads = [sum(lengths[i] for i in p) for p in parts]
                assert max(loads) <= 4 * sum(lengths) / (3 * world) + max(lengths) / 3 + 1
    with pytest.raises(ValueError):
        shard_indices(3, 0, 2, lengths=[1, 2])
