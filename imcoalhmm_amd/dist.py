"""Multi-GPU sharding of the likelihood: one process per GPU, chunks partitioned statically,
one RCCL sum of the per-rank partial log-likelihoods (backend "nccl" is RCCL on ROCm).

The path shards exactly where the reference's ``Likelihood.__call__`` sums over forwarders
(src/IMCoalHMM/likelihood.py:33): chunks are independent, every rank holds the (tiny) parameter
set, and the only exchange is ``all_reduce(sum)`` of B doubles per evaluation batch.

Import order: ``import torch`` (and ``torch.cuda.set_device``) BEFORE the first call into libimcoal_fwd - the PyTorch
wheel bundles its own HIP runtime, and whichever runtime initialises the GPU first is the only one that sees it
(bench.py does exactly this; the other way round torch reports "No HIP GPUs are available").
"""
import ctypes

import numpy as np

from . import _capi, hmm
from .likelihood import build_hmms


def shard_indices(n_chunks, rank, world_size, lengths=None):
    """Static partition of chunk indices over ranks.

    Without ``lengths``: round-robin, chunk i -> rank i mod world_size (right for equal-length chunks).
    With ``lengths`` (columns per chunk): balanced by column count - chunks are dealt longest first, each to the rank
    that so far holds the fewest columns (ties: the lowest rank), the classic LPT rule: no rank ends up with more than
    4/3 of the optimum.  Deterministic, so every rank computes the same partition without communicating.  The
    returned indices are in ascending order."""
    if lengths is None:
        return list(range(rank, n_chunks, world_size))
    if len(lengths) != n_chunks:
        raise ValueError("lengths must have one entry per chunk")
    load = [0] * world_size
    mine = []
    for i in sorted(range(n_chunks), key=lambda k: (-int(lengths[k]), k)):
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += int(lengths[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


class DistributedLikelihood(object):
    """``Likelihood`` whose forwarders are this rank's shard; every rank gets the global value.

    ``local_eval(pis, Ts, Es) -> tensor[B]`` may be injected (CPU/gloo tests); by default the HIP
    library writes the rank's partial sums straight into a device tensor on torch's current stream
    and RCCL reduces it in place.
    """

    def __init__(self, model, local_forwarders, group=None, device=None, local_eval=None, reduce_on_host=False,
                 reduction="allreduce", recompress=True, force_collective=False):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.model = model
        if hasattr(local_forwarders, '__iter__'):
            self.forwarders = list(local_forwarders)
        else:
            self.forwarders = [local_forwarders]
        self.group = group
        self.device = device
        self._local_eval = local_eval or self._hip_eval
        self.reduce_on_host = reduce_on_host     # gloo rehearsals: reduce a host copy of the partial sums
        # "allreduce": one all_reduce(sum) of B doubles (the order of the additions is the backend's);
        # "ordered":   one all_gather of the per-rank partial sums, added on the host in rank order 0, 1, 2, ... from
        #              0.0 - every rank gets the same bits whatever the collective's algorithm or topology (SURVEY.md
        #              section 5: fixed reduction order for run-to-run reproducibility)
        if reduction not in ("allreduce", "ordered"):
            raise ValueError("reduction must be 'allreduce' or 'ordered'")
        self.reduction = reduction
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        # a single rank has nothing to reduce: the library's synchronous entry point writes the results straight into
        # host memory (no device tensor, no reduction kernel, no device-to-host copy through torch)
        # (force_collective=True keeps the device-output entry point and the collective even for one rank: how a
        # single GPU exercises exactly the code N ranks run - RCCL initialisation, imc_forward_batch_device on torch's
        # stream, the in-place all_reduce / the all_gather - tests/test_gpu_nccl.py)
        self.force_collective = bool(force_collective)
        self._direct = local_eval is None and self.world_size == 1 and not reduce_on_host and not self.force_collective
        self._harr = None
        if recompress and local_eval is None:   # this rank's shard: one pair dictionary trained on all of its chunks
            hmm.recompress(self.forwarders)

    def _hip_eval(self, pis, Ts, Es):
        torch = self._torch
        dev = self.device if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        B, n = pis.shape
        partial = torch.empty(B, dtype=torch.float64, device=dev)
        handles = [f.handle for f in self.forwarders]
        stream = torch.cuda.current_stream(dev).cuda_stream
        _capi.check(_capi.lib().imc_forward_batch_device(
            _capi.handle_array(handles), len(handles), B, n, Es.shape[2],
            _capi.dptr(pis), _capi.dptr(Ts), _capi.dptr(Es),
            ctypes.c_void_p(partial.data_ptr()), ctypes.c_void_p(stream)))
        return partial

    def forward_params_batch(self, pis, Ts, Es, reduce=True):
        """Global log-likelihoods (float64[B]) for B parameter sets; collective over the group.
        ``reduce=False``: this rank's partial sums only, through the same device-output entry point but without the
        collective (bench.py times a rank's shard alone this way)."""
        pis, Ts, Es = hmm._batch_params(pis, Ts, Es)
        if self._direct:
            return hmm.forward_chunks_batch([f.handle for f in self.forwarders], pis, Ts, Es)
        partial = self._local_eval(pis, Ts, Es)
        if self.reduce_on_host:
            partial = partial.cpu()
        collective = reduce and self._dist.is_initialized() and (self.world_size > 1 or self.force_collective)
        if collective and self.reduction == "ordered":
            parts = [self._torch.empty_like(partial) for _ in range(self.world_size)]
            self._dist.all_gather(parts, partial, group=self.group)
            total = np.zeros(partial.shape[0], dtype=np.float64)
            for part in parts:                          # rank order, left to right from 0.0
                total = total + part.detach().cpu().numpy()
            return total
        if collective:
            self._dist.all_reduce(partial, op=self._dist.ReduceOp.SUM, group=self.group)
        return partial.detach().cpu().numpy()

    def forward_params(self, pi, T, E, reduce=True):
        pi, T, E = hmm._params(pi, T, E)
        if self._direct:
            if self._harr is None:
                self._harr = _capi.handle_array([f.handle for f in self.forwarders])
            return _capi.forward1(self._harr, len(self.forwarders), pi, T, E)
        return float(self.forward_params_batch(pi[None], T[None], E[None], reduce=reduce)[0])

    def __call__(self, *parameters):
        if not self.model.valid_parameters(*parameters):
            return -float('inf')
        return self.forward_params(*self.model.build_hidden_markov_model(*parameters))

    def batch(self, thetas):
        thetas = [np.asarray(t, dtype=np.float64) for t in thetas]
        out = np.full(len(thetas), -np.inf, dtype=np.float64)
        valid = [k for k, t in enumerate(thetas) if self.model.valid_parameters(t)]
        if valid:
            out[valid] = self.forward_params_batch(*build_hmms(self.model, [thetas[k] for k in valid]))
        return out


def slice_bounds(n_columns, rank, world_size):
    """Contiguous column range [begin, end) of rank's slice of one long alignment (balanced to one column)."""
    return (n_columns * rank) // world_size, (n_columns * (rank + 1)) // world_size


class SplitAlignmentLikelihood(object):
    """ONE long alignment across GPUs: rank r holds columns ``slice_bounds(L, r, world)``
    as its own Forwarder; rank 0 exports the forward vector after its slice, every other rank its slice's exact
    N x N transfer operator (``imc_forward_state``); one ``all_gather`` of N*N + N doubles per parameter set and
    a tiny ordered combine give every rank the log-likelihood of the whole alignment.

    ``local_state(pis, Ts, Es, as_operator) -> (values, exponents)`` may be injected (CPU/gloo tests).
    """

    def __init__(self, model, local_forwarder, group=None, local_state=None, gather_device=None, force_collective=False):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.model = model
        self.forwarder = local_forwarder
        self.group = group
        self.gather_device = gather_device       # None: host tensors (gloo); a cuda device for RCCL
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._local_state = local_state or self._hip_state
        self.force_collective = bool(force_collective)   # one rank still goes through all_gather (tests/test_gpu_nccl.py)

    def _hip_state(self, pis, Ts, Es, as_operator):
        values, exps = hmm.forward_states([self.forwarder.handle], pis, Ts, Es, as_operator)
        return values[:, 0], exps[:, 0]

    def forward_params_batch(self, pis, Ts, Es):
        torch = self._torch
        pis, Ts, Es = hmm._batch_params(pis, Ts, Es)
        B, n = pis.shape
        values, exps = self._local_state(pis, Ts, Es, self.rank > 0)
        # one fixed-size record per rank: N*N values (the vector sits in the first N) + N exponents
        rec = np.zeros((B, n * n + n), dtype=np.float64)
        if self.rank > 0:
            rec[:, :n * n] = np.asarray(values, dtype=np.float64).reshape(B, n * n)
            rec[:, n * n:] = np.asarray(exps, dtype=np.float64).reshape(B, n)
        else:
            rec[:, :n] = np.asarray(values, dtype=np.float64).reshape(B, n)
            rec[:, n * n] = np.asarray(exps, dtype=np.float64).reshape(B)
        mine = torch.from_numpy(rec)
        if self.gather_device is not None:
            mine = mine.to(self.gather_device)
        if self.world_size > 1 or (self.force_collective and self._dist.is_initialized()):
            parts = [torch.empty_like(mine) for _ in range(self.world_size)]
            self._dist.all_gather(parts, mine, group=self.group)
        else:
            parts = [mine]
        parts = [p.cpu().numpy() for p in parts]
        out = np.empty(B, dtype=np.float64)
        for b in range(B):
            ops = [parts[r][b, :n * n].reshape(n, n) for r in range(1, self.world_size)]
            opx = [parts[r][b, n * n:].astype(np.int64) for r in range(1, self.world_size)]
            out[b] = hmm.combine_states(parts[0][b, :n], int(parts[0][b, n * n]), ops, opx)
        return out

    def forward_params(self, pi, T, E):
        pi, T, E = hmm._params(pi, T, E)
        return float(self.forward_params_batch(pi[None], T[None], E[None])[0])

    def __call__(self, *parameters):
        if not self.model.valid_parameters(*parameters):
            return -float('inf')
        return self.forward_params(*self.model.build_hidden_markov_model(*parameters))


class ProposalShardedLikelihood(object):
    """The second sharding axis of SURVEY.md section 8e: when chunks are few (one long alignment) but proposals are
    many, every rank holds ALL chunks and evaluates every ``world_size``-th parameter set; one ``all_gather`` of the
    per-rank values gives every rank the whole batch.  ``local_eval(pis, Ts, Es) -> float64[b]`` may be injected
    (CPU/gloo tests); by default it is ``forward_chunks_batch`` over this process's forwarders.
    """

    def __init__(self, model, forwarders, group=None, local_eval=None, gather_device=None, force_collective=False):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.model = model
        self.forwarders = list(forwarders) if hasattr(forwarders, '__iter__') else [forwarders]
        self.group = group
        self.gather_device = gather_device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._local_eval = local_eval or (lambda pis, Ts, Es: hmm.forward_chunks_batch(
            [f.handle for f in self.forwarders], pis, Ts, Es))
        self.force_collective = bool(force_collective)   # one rank still goes through all_gather (tests/test_gpu_nccl.py)

    def forward_params_batch(self, pis, Ts, Es):
        torch = self._torch
        pis, Ts, Es = hmm._batch_params(pis, Ts, Es)
        B = pis.shape[0]
        mine = list(range(self.rank, B, self.world_size))
        per_rank = (B + self.world_size - 1) // self.world_size
        vals = np.zeros(per_rank, dtype=np.float64)                 # fixed-size record, padded with zeros
        if mine:
            vals[:len(mine)] = np.asarray(self._local_eval(pis[mine], Ts[mine], Es[mine]), dtype=np.float64)
        t = torch.from_numpy(vals)
        if self.gather_device is not None:
            t = t.to(self.gather_device)
        if self.world_size > 1 or (self.force_collective and self._dist.is_initialized()):
            parts = [torch.empty_like(t) for _ in range(self.world_size)]
            self._dist.all_gather(parts, t, group=self.group)
        else:
            parts = [t]
        out = np.empty(B, dtype=np.float64)
        for r, part in enumerate(parts):
            idx = list(range(r, B, self.world_size))
            out[idx] = part.cpu().numpy()[:len(idx)]
        return out

    def batch(self, thetas):
        thetas = [np.asarray(t, dtype=np.float64) for t in thetas]
        out = np.full(len(thetas), -np.inf, dtype=np.float64)
        valid = [k for k, t in enumerate(thetas) if self.model.valid_parameters(t)]
        if valid:                       # every rank builds every HMM (cheap, deterministic) and evaluates its share
            out[valid] = self.forward_params_batch(*build_hmms(self.model, [thetas[k] for k in valid]))
        return out

    def __call__(self, *parameters):
        if not self.model.valid_parameters(*parameters):
            return -float('inf')
        pi, T, E = hmm._params(*self.model.build_hidden_markov_model(*parameters))
        return float(self.forward_params_batch(pi[None], T[None], E[None])[0])
