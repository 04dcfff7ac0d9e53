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
