"""The multi-GPU code path on ONE GPU: a world-size-1 ``nccl`` (= RCCL) process group.

`DistributedLikelihood` takes a direct path when there is nothing to reduce (dist.py: `_direct`), so on a one-GPU box the
code N ranks actually run - RCCL initialisation, `imc_forward_batch_device` writing the rank's partial sums into a torch
tensor on torch's stream, the in-place `all_reduce` / the `all_gather` of the ordered reduction - never executed
(the 2-rank rehearsals reduce over gloo on the host).  `force_collective=True` keeps all of it for a single rank.  The
shard point is the sum over forwarders at /root/reference/src/IMCoalHMM/likelihood.py:33.

Runs in a child process: torch must be imported, and the device selected, before the first library call (dist.py)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import os, socket, sys
    import numpy as np
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% port, rank=0, world_size=1, device_id=dev)

    from imcoalhmm_amd import Forwarder, _capi, models, synth
    from imcoalhmm_amd.dist import DistributedLikelihood, ProposalShardedLikelihood, SplitAlignmentLikelihood
    from imcoalhmm_amd.hmm import forward_chunks_batch
    L = _capi.lib()
    _capi.check(L.imc_set_device(0))

    def proposals(model, theta0, B, seed):
        rng = np.random.default_rng(seed)
        thetas = np.asarray(theta0) * np.exp(0.1 * rng.standard_normal((B, len(theta0))))
        thetas[0] = theta0
        return thetas, model.build_batch(thetas)

    for n_states, lens in ((20, [300_000, 70_001, 0, 5_000, 1_200_000]), (10, [65_255])):
        model = models.IsolationModel(n_states)
        theta0 = (0.001, 1000.0, 0.4)
        pi, T, E = model.build_hidden_markov_model(np.array(theta0))
        chunks = [synth.sample_alignment(pi, T, E, m, seed=900 + k) for k, m in enumerate(lens)]
        fw = [Forwarder.from_array(c, 3) for c in chunks]
        handles = [f.handle for f in fw]
        direct = DistributedLikelihood(model, fw, device=dev)
        assert direct._direct
        for B in (1, 64):
            thetas, (pis, Ts, Es) = proposals(model, theta0, B, seed=B)
            want = forward_chunks_batch(handles, pis, Ts, Es)                  # the synchronous entry point
            for reduction in ("allreduce", "ordered"):
                ll = DistributedLikelihood(model, fw, device=dev, reduction=reduction, force_collective=True)
                assert not ll._direct
                got = ll.forward_params_batch(pis, Ts, Es)                     # device output + RCCL collective
                assert got.dtype == np.float64 and got.shape == (B,)
                assert np.array_equal(got, want), (n_states, B, reduction, got[:3], want[:3])
                again = ll.batch(list(thetas))                                 # through the model layer, twice in a row
                assert np.array_equal(again, want)
                assert ll(np.array(theta0)) == direct(np.array(theta0))      # (both through the B = 1 plan)
            # on a side stream: the library must order itself on torch's CURRENT stream
            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                ll = DistributedLikelihood(model, fw, device=dev, force_collective=True)
                assert np.array_equal(ll.forward_params_batch(pis, Ts, Es), want)
        # invalid parameters never reach the device
        assert ll(np.array((-1.0, 1000.0, 0.4))) == -float("inf")

        # proposal shards: every rank holds all chunks, all_gather of the values on the device
        thetas, (pis, Ts, Es) = proposals(model, theta0, 7, seed=3)
        want = forward_chunks_batch(handles, pis, Ts, Es)
        ps = ProposalShardedLikelihood(model, fw, gather_device=dev, force_collective=True)
        assert np.array_equal(ps.forward_params_batch(pis, Ts, Es), want)
        assert np.array_equal(ps.batch(list(thetas)), want)

        # one alignment "split" over one rank: state export + all_gather on the device + the ordered combine
        long_fw = max(fw, key=len)
        want1 = forward_chunks_batch([long_fw.handle], pis, Ts, Es)
        sp = SplitAlignmentLikelihood(model, long_fw, gather_device=dev, force_collective=True)
        got1 = sp.forward_params_batch(pis, Ts, Es)
        assert np.all(np.abs(got1 - want1) <= 1e-12 * np.abs(want1)), (got1, want1)
        del fw, long_fw, ll, ps, sp, direct
        print("states %%d ok" %% n_states, flush=True)

    dist.barrier()
    dist.destroy_process_group()
    print("nccl world-1 ok", flush=True)
''') % (REPO,)


def test_world_size_one_nccl_group_runs_the_collective_path(tmp_path):
    script = tmp_path / "nccl_world1.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and "nccl world-1 ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
