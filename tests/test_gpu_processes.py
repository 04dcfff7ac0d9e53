"""The reference's MC3 builds its Forwarders inside multiprocessing children (mcmc.py:99-145: one OS process per
chain, `_set_chain()` constructs Forwarders + Likelihood in the child, results come back through Queues).  The
library touches HIP lazily and per PID, so that pattern must work: a parent that never used the GPU forks two
chain processes, each builds its own Forwarder and evaluates."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import multiprocessing as mp, os, sys
    import numpy as np
    sys.path.insert(0, %r)

    def chain(k, q, path):
        from imcoalhmm_amd import Forwarder, Likelihood            # first HIP use happens here, in the child
        d = np.load(os.path.join(%r, "tests", "golden", "hmm_params.npz"))
        pi, T, E = d["iso10_t%%d_pi" %% k], d["iso10_t%%d_T" %% k], d["iso10_t%%d_E" %% k]
        class M(object):
            def valid_parameters(self, p): return True
            def build_hidden_markov_model(self, p): return pi, T, E
        ll = Likelihood(M(), [Forwarder(path, NSYM=3)])
        q.put((k, ll(np.array([1.0]))))

    if __name__ == "__main__":
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        ps = [ctx.Process(target=chain, args=(k, q, sys.argv[1])) for k in range(2)]
        for p in ps: p.start()
        res = dict(q.get(timeout=120) for _ in ps)
        for p in ps:
            p.join(60)
            assert p.exitcode == 0
        print(repr(res[0]), repr(res[1]))
''') % (REPO, REPO)


def test_forwarders_built_in_forked_children(tmp_path, oracle, hmm_params, example_pairs):
    obs = example_pairs["hg18__pantro2"]
    path = tmp_path / "pair.txt"
    path.write_text(" ".join(str(int(s)) for s in obs) + " ")
    script = tmp_path / "chains.py"
    script.write_text(SCRIPT)
    out = subprocess.run([sys.executable, str(script), str(path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    a, b = (float(x) for x in out.stdout.split()[-2:])
    for got, key in ((a, "iso10_t0"), (b, "iso10_t1")):
        want = oracle.forward_scaled(*hmm_params(key), obs)
        assert abs(got - want) / abs(want) < 1e-11


FORK_AFTER_INIT = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from imcoalhmm_amd import Forwarder, _capi
    d = np.load(os.path.join(%r, "tests", "golden", "hmm_params.npz"))
    pi, T, E = d["iso10_t0_pi"], d["iso10_t0_T"], d["iso10_t0_E"]
    obs = (np.arange(5000) %% 3).astype(np.uint8)
    f = Forwarder.from_array(obs, 3)
    before = f.forward(pi, T, E)                     # the parent owns a live context, a cached plan, device buffers
    pid = os.fork()
    if pid == 0:                                     # child: every library call must be refused, nothing may touch HIP
        code = 1
        try:
            Forwarder.from_array(obs, 3)
        except _capi.ImcError as e:
            code = 0 if "before fork" in str(e) else 2
        try:
            f.forward(pi, T, E)
            code = 3
        except _capi.ImcError as e:
            code = code if "before fork" in str(e) else 4
        os._exit(code)                               # (no interpreter teardown of inherited handles)
    _, status = os.waitpid(pid, 0)
    after = f.forward(pi, T, E)                      # the parent is unaffected
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    assert before == after
    print("ok")
''') % (REPO, REPO)


def test_fork_after_initialisation_is_refused(tmp_path):
    """A child forked after the parent has used the GPU cannot use HIP (its state does not survive fork): every
    library call in it must fail with a clear IMC_ERR_HIP instead of re-initialising or freeing the parent's device
    pointers, and the parent must carry on unharmed.  (Reference pattern: mcmc.py:112-121 forks BEFORE building.)"""
    script = tmp_path / "fork_after_init.py"
    script.write_text(FORK_AFTER_INIT)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2000:])


def test_concurrent_calls_from_threads(oracle, hmm_params, example_pairs):
    """The Python shim releases the GIL inside the C call; the library serialises the ENQUEUE of calls with one mutex (and
    two callers of the same plan altogether).  Four threads hammering different Forwarders / models must each get their
    own correct value."""
    from concurrent.futures import ThreadPoolExecutor
    from imcoalhmm_amd import Forwarder
    jobs = []
    for k, (pname, mkey) in enumerate([("hg18__pantro2", "iso10_t0"), ("hg18__bonobo", "iso20_t0"),
                                       ("pantro2__bonobo", "im20_t0"), ("hg18__ponabe2", "iso20_t2")]):
        obs = example_pairs[pname]
        jobs.append((Forwarder.from_array(obs, 3), hmm_params(mkey), oracle.forward_scaled(*hmm_params(mkey), obs)))

    def work(j):
        f, (pi, T, E), want = jobs[j % 4]
        vals = [f.forward(pi, T, E) for _ in range(25)]
        return max(abs(v - want) / abs(want) for v in vals), len(set(vals))

    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(work, range(8)))
    assert all(err < 1e-11 and distinct == 1 for err, distinct in res), res


def test_threads_enqueue_while_another_waits(oracle, hmm_params):
    """A synchronous call holds the library's mutex while it enqueues, not while it waits for its results: threads with
    their own Forwarders must be able to queue their evaluations behind one another (round 2 serialised them completely,
    host turn-around included).  Checked: every value is right and bit-identical call to call; a thread that shares a
    Forwarder - and therefore a plan and its result slots - with another still gets its own values; chunks are freed
    while other threads evaluate; and four threads together are not slower than one thread doing all the calls."""
    import threading
    import time
    from imcoalhmm_amd import Forwarder, synth
    pi, T, E = hmm_params("iso20_t0")
    hm2 = hmm_params("iso20_t1")
    chunks = [synth.sample_alignment(pi, T, E, 1_500_000 + 100_000 * k, seed=400 + k) for k in range(4)]
    fws = [Forwarder.from_array(c, 3) for c in chunks]
    want = [(oracle.forward_scaled(pi, T, E, c), oracle.forward_scaled(*hm2, c)) for c in chunks]
    reps = 150

    def work(k, out):
        vals = []
        for r in range(reps):
            vals.append((fws[k].forward(pi, T, E), fws[k].forward(*hm2)))
            if r % 50 == 7:                       # create / evaluate / free a small chunk in the middle of it all
                g = Forwarder.from_array(chunks[k][:30_000], 3)
                vals.append((g.forward(pi, T, E), None))
                del g
                vals.pop()
        out[k] = vals

    for k in range(4):
        work(k, {})                               # plans built, caches warm
    t0 = time.perf_counter()
    seq = {}
    for k in range(4):
        work(k, seq)
    t_seq = time.perf_counter() - t0
    par = {}
    threads = [threading.Thread(target=work, args=(k, par)) for k in range(4)] + [threading.Thread(target=work, args=(0, {}))]
    t0 = time.perf_counter()
    for th in threads[:4]:
        th.start()
    for th in threads[:4]:
        th.join()
    t_par = time.perf_counter() - t0
    shared = {}
    pair = [threading.Thread(target=work, args=(0, shared)), threading.Thread(target=lambda: work(0, {}))]   # two threads, ONE plan
    for th in pair:
        th.start()
    for th in pair:
        th.join()
    for res in (seq, par, shared):
        for k, vals in res.items():
            assert len(set(vals)) == 1, (k, len(set(vals)))
            assert abs(vals[0][0] - want[k][0]) <= 1e-11 * abs(want[k][0]) and abs(vals[0][1] - want[k][1]) <= 1e-11 * abs(want[k][1])
    print("threads: sequential %.1f ms, four threads %.1f ms" % (t_seq * 1e3, t_par * 1e3))
    assert t_par < 1.5 * t_seq, (t_seq, t_par)          # (no slower than serialised, with room for a noisy box: 1.10 failed once under IMC_GUARD=1)


def test_integration_md_stub_is_valid(tmp_path, oracle, hmm_params, example_pairs):
    """The ctypes stub INTEGRATION.md proposes for src/IMCoalHMM/hmm.py must actually work against the library."""
    import re
    from imcoalhmm_amd import _capi
    md = open(os.path.join(REPO, "INTEGRATION.md")).read()
    code = re.search(r"## Option B.*?```python\n(.*?)```", md, flags=re.S).group(1)
    os.environ["IMCOAL_FWD_LIB"] = _capi.LIB_PATH
    ns = {}
    exec(compile(code, "INTEGRATION.md:OptionB", "exec"), ns)
    obs = example_pairs["hg18__pantro2"]
    path = tmp_path / "pair.txt"
    path.write_text(" ".join(str(int(s)) for s in obs) + " ")
    f = ns["Forwarder"](str(path), NSYM=3)
    pi, T, E = hmm_params("iso10_t0")
    got = f.forward(np.matrix(pi), np.matrix(T), np.matrix(E))      # the reference's model layer returns numpy.matrix
    want = oracle.forward_scaled(pi, T, E, obs)
    assert abs(got - want) / abs(want) < 1e-11


DEVICE_PATH = textwrap.dedent('''
    import ctypes, os, sys
    import numpy as np
    import torch                                   # torch first: its bundled HIP runtime must be the one that initialises
    sys.path.insert(0, %r)
    from imcoalhmm_amd import Forwarder, _capi, synth
    from imcoalhmm_amd.hmm import forward_chunks_batch
    from oracle import oracle_lib
    oracle_lib.build()
    d = np.load(os.path.join(%r, "tests", "golden", "hmm_params.npz"))
    P = lambda k: (d[k + "_pi"], d[k + "_T"], d[k + "_E"])
    torch.cuda.set_device(0)
    chunks = [synth.sample_alignment(*P("iso20_t0"), n, seed=900 + k) for k, n in enumerate((70_000, 5_000, 33, 0, 120_000))]
    fw = [Forwarder.from_array(c, 3) for c in chunks]
    handles = _capi.handle_array([f.handle for f in fw])
    keys = ["iso20_t0", "iso20_t1", "iso20_t2", "iso20_t3", "im20_t0"]
    sets = []
    for a, b in zip(keys, keys[1:] + keys[:1]):          # B = 2 parameter sets per call
        pa, pb = P(a), P(b)
        sets.append(tuple(np.ascontiguousarray(np.stack([pa[k], pb[k]])) for k in range(3)))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    outs = [torch.full((2,), float("nan"), dtype=torch.float64, device=dev) for _ in sets]
    L = _capi.lib()
    for (pis, Ts, Es), out in zip(sets, outs):           # five calls back to back, no synchronisation in between
        _capi.check(L.imc_forward_batch_device(handles, len(fw), 2, 20, 3, _capi.dptr(pis), _capi.dptr(Ts), _capi.dptr(Es),
                                               ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
    torch.cuda.synchronize()
    for (pis, Ts, Es), out in zip(sets, outs):
        want = forward_chunks_batch([f.handle for f in fw], pis, Ts, Es)
        got = out.cpu().numpy()
        assert np.array_equal(got, want), (got, want)      # same plan, same kernels, same left-to-right chunk sum
        for b in range(2):
            ref = sum(oracle_lib.forward_scaled(pis[b], Ts[b], Es[b], c) for c in chunks)
            assert abs(got[b] - ref) / abs(ref) < 1e-11
    # Stream ordering (round-2 bug: a NULL stream handle used to mean the library's own non-blocking stream, so work
    # queued for torch's DEFAULT stream - handle 0 - was not ordered before torch's next use of the output tensor).
    # A long chunk (the evaluation takes ~1 ms), five different parameter sets, every result read back through torch's
    # own stream ordering ONLY: a stale or unfinished buffer cannot equal the synchronous value of its own set.
    big = [Forwarder.from_array(synth.sample_alignment(*P("iso20_t0"), 30_000_000, seed=77), 3)]
    hb = _capi.handle_array([f.handle for f in big])
    want_big = [forward_chunks_batch([big[0].handle], *st) for st in sets]
    assert stream == 0 or stream is None                 # torch's current stream here IS the default stream
    out = torch.full((2,), float("nan"), dtype=torch.float64, device=dev)
    for (pis, Ts, Es), want in zip(sets, want_big):
        _capi.check(L.imc_forward_batch_device(hb, 1, 2, 20, 3, _capi.dptr(pis), _capi.dptr(Ts), _capi.dptr(Es),
                                               ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)))
        got = out.cpu().numpy()                          # no torch.cuda.synchronize(): the copy is ordered on the stream
        assert np.array_equal(got, want), (got, want)
    # ... on a side stream too, and alternating with the synchronous entry point (the plan's buffers are shared: a call
    # on another stream than its predecessor's waits for it)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for (pis, Ts, Es), want in zip(sets, want_big):
            _capi.check(L.imc_forward_batch_device(hb, 1, 2, 20, 3, _capi.dptr(pis), _capi.dptr(Ts), _capi.dptr(Es),
                                                   ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(side.cuda_stream)))
            again = forward_chunks_batch([big[0].handle], pis, Ts, Es)      # library stream, right behind the side stream
            got = out.cpu().numpy()
            assert np.array_equal(got, want) and np.array_equal(again, want), (got, again, want)
    print("ok")
''') % (REPO, REPO)


def test_device_output_path_is_asynchronous_and_exact(tmp_path):
    """imc_forward_batch_device (the multi-GPU building block, dist.py): partial sums stay on the device, the call
    returns after enqueueing and stages its parameters through two pinned slots - five calls with different parameter
    sets are issued back to back on torch's stream without any synchronisation, then every result must equal the
    synchronous path's bit for bit.  (Own process: torch has to be imported before the library first touches HIP -
    the wheel bundles its own HIP runtime - which is also the order bench.py and dist.py users follow.)"""
    script = tmp_path / "device_path.py"
    script.write_text(DEVICE_PATH)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2000:])


AB_SWITCHES = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %r)
    from imcoalhmm_amd import Forwarder, _capi, synth
    from imcoalhmm_amd.hmm import forward_chunks_batch
    L = _capi.lib()
    out = []
    tails = []
    # register-blocked MFMA kernel on a global table (two dictionary depths per table launch, or one)
    _capi.check(L.imc_set_compression(3)); _capi.check(L.imc_set_blocked_kernel(5))
    for n in (10, 20):
        hmms = [synth.random_hmm(n, 3, seed=50 + n + b, stay=0.995) for b in range(2)]
        chunks = [synth.sample_alignment(*hmms[0], m, seed=9 + k) for k, m in enumerate((900_000, 4099, 120_000))]
        fw = [Forwarder.from_array(c, 3) for c in chunks]
        v = forward_chunks_batch([f.handle for f in fw], *(np.stack([h[k] for h in hmms]) for k in range(3)), per_chunk=True)
        assert "k_zpropagate4" in _capi.last_plan()["kernels"]
        out += [float(x).hex() for x in v.ravel()]
        _capi.check(L.imc_dictionary_reset())
    # small launches of the LDS-table kernel: <= 32 workgroups (each fetches the parameter set itself, or k_stage_params
    # does), and chunks of at most 32 workgroups (the chunk's last workgroup can finish the chunk: IMC_FUSE_TAIL)
    _capi.check(L.imc_set_compression(1)); _capi.check(L.imc_set_blocked_kernel(4))
    d = np.load(os.path.join(%r, "tests", "golden", "hmm_params.npz"))
    for n in (10, 20):
        hmms = [tuple(d["iso%%d_t%%d_%%s" %% (n, b, k)] for k in ("pi", "T", "E")) for b in range(2)]
        f = Forwarder.from_array(synth.sample_alignment(*hmms[0], 60_000, seed=n), 3)
        out.append(float(f.forward(*hmms[0])).hex())
        assert "k_zpropagate3" in _capi.last_plan()["kernels"]
        out.append(_capi.last_plan()["kernels"].replace("+fused-tail", ""))
        tails.append("fused-tail" in _capi.last_plan()["kernels"])
        fw = [f] + [Forwarder.from_array(synth.sample_alignment(*hmms[0], m, seed=n + k), 3) for k, m in enumerate((40_000, 5_000, 33_000))]
        v = forward_chunks_batch([g.handle for g in fw], *(np.stack([h[k] for h in hmms]) for k in range(3)), per_chunk=True)
        out += [float(x).hex() for x in v.ravel()]
        tails.append("fused-tail" in _capi.last_plan()["kernels"])
        del f, fw
        _capi.check(L.imc_dictionary_reset())
    # mat-vec chain at 150 states (packed or padded operator table), and the hand-off tails of a long chunk
    _capi.check(L.imc_set_compression(1)); _capi.check(L.imc_set_blocked_kernel(4))
    h150 = synth.random_hmm(150, 3, seed=7, stay=0.5)
    many = [Forwarder.from_array(synth.sample_alignment(*h150, 6000 + 37 * k, seed=k), 3) for k in range(40)]
    v = forward_chunks_batch([f.handle for f in many], *(x[None] for x in h150), per_chunk=True)
    out += [float(x).hex() for x in v.ravel()]
    out.append(_capi.last_plan()["kernels"])
    long1 = Forwarder.from_array(synth.sample_alignment(*h150, 1_500_000, seed=3), 3)
    out.append(float(long1.forward(*h150)).hex())
    out.append(_capi.last_plan()["kernels"])
    print("TAILS", "".join("1" if x else "0" for x in tails))
    print(" ".join(out))
''') % (REPO, REPO)


def test_ab_switches_change_nothing_but_the_schedule(tmp_path):
    """The launch-schedule switches of the header (IMC_TABLE_PAIRS: two dictionary depths per table launch with the
    second depth's children recomputed; IMC_PACK_TABLE: the mat-vec chain reads a packed copy of the operator table;
    IMC_FUSE_HEAD: the first table launch fetches the parameters and builds the raw operators itself; IMC_TABLE_TRIPLES:
    three depths per table launch, one wavefront per token)
    re-associate nothing: every value must come out bit for bit the same with the switch on and off.  (Own processes:
    the switches are read when the library's context is created.)"""
    script = tmp_path / "ab.py"
    script.write_text(AB_SWITCHES)
    outs = {}
    combos = ((1, 1, 1, 1), (0, 1, 1, 1), (1, 0, 1, 1), (1, 1, 0, 1), (1, 1, 1, 0), (1, 1, 0, 0))
    for pairs, pack, fuse, triples in combos:
        env = dict(os.environ, IMC_TABLE_PAIRS=str(pairs), IMC_PACK_TABLE=str(pack), IMC_FUSE_HEAD=str(fuse),
                   IMC_TABLE_TRIPLES=str(triples))
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
        outs[(pairs, pack, fuse, triples)] = r.stdout.strip().splitlines()[-1]
    assert len(set(outs.values())) == 1, outs
    assert "k_big_vector" in outs[combos[0]] and "k_big_propagate" in outs[combos[0]]      # both large-N paths were on the route
    # IMC_XCD_AFFINE=0: k_zpropagate4 on the plain (blocks, B) grid instead of the XCD-affine one - placement only
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=dict(os.environ, IMC_XCD_AFFINE="0"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1] == outs[combos[0]]
    # IMC_FUSE_TAIL (the chunk's last workgroup finishes the chunk instead of k_chain launches; 2 = wherever a chunk is at
    # most 32 workgroups, 0 = never) re-associates the last few products and the final sum: same values to 1e-13, not
    # the same bits
    alt, used = {}, {}
    for mode in ("0", "2"):
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=dict(os.environ, IMC_FUSE_TAIL=mode))
        assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
        alt[mode] = r.stdout.strip().splitlines()[-1].split()
        used[mode] = [l for l in r.stdout.splitlines() if l.startswith("TAILS")][-1].split()[1]
    assert used["0"] == "0000" and used["2"] == "1111", used     # (the two routes really are different)
    for other in alt.values():
        for x, y in zip(outs[combos[0]].split(), other):
            if x.startswith(("0x", "-0x")):
                fx, fy = float.fromhex(x), float.fromhex(y)
                assert abs(fx - fy) <= 1e-13 * abs(fx), (x, y)
            else:
                assert x == y
