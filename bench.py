#!/usr/bin/env python3
"""bench.py - alignment columns/s of the MI355X forward log-likelihood path (BASELINE.json metric).

A "step" is one evaluation of the summed log-likelihood (one pass of the hot path) over this
job's synthetic alignment chunks, which are resident in HBM before the timed region starts; the
per-step host input is only (pi, T, E) (a few KB), exactly what Forwarder.forward receives.

  --gpus 1 : BASELINE config[1] - isolation model, 20 states, 1 x 100 Mbp synthetic pairwise alignment
             (--workload config4-slice gives the per-GPU slice of the multi-GPU workload on one GPU)
  --gpus N : BASELINE config[3] sliced per GPU - 32 x 10 Mbp chunks per rank (256 chunks at N=8, weak scaling),
             chunks sharded statically, one RCCL all-reduce(sum) of the partial log-likelihoods per step.
             The line carries its OWN single-GPU base: before the group run every rank times its shard alone through
             the same entry point without the collective (`single_gpu_same_workload`, `scaling_efficiency`), and the
             per-rank step times (`rank_ms_per_step`), so the N>1 lines can be judged without the N=1 line (whose
             workload, config[1], is a different one).
             --strong: BASELINE config[3] as worded - 256 chunks in total, 256/N per rank (also valid at N=1).
             Launched by the driver under torch.distributed.run; run directly (no WORLD_SIZE in the
             environment) it starts its own N rank processes before anything touches the GPU.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` and, at N=1,
`cpu_baseline` and `extra_configs` (the other BASELINE configs at their per-GPU size, a few steps each).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 fp64 FMA lanes x 2 x 2.4 GHz; the fp64 MFMA peak is the same
PIECE = 10_000_000             # synthetic alignments are generated in pieces of this many columns
GEN_WORKERS = 0                # --gen-workers (0 = automatic)
CONDITION_MS = 60.0            # --condition-ms: untimed evaluations before the warmup steps (timed_steps)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--states", type=int, default=20)
    ap.add_argument("--fixture", type=str, default="",
                    help="(pi,T,E) fixture key in tests/golden/hmm_params.npz, e.g. im150_t0 (default iso<states>_t0)")
    ap.add_argument("--workload", choices=("auto", "config2", "config4-slice"), default="auto",
                    help="auto: config2 (1 x 1e8 columns) on one GPU, config4-slice (32 x 1e7 per rank) on several")
    ap.add_argument("--columns", type=int, default=0, help="override columns per chunk")
    ap.add_argument("--chunks", type=int, default=0, help="override chunks per rank")
    ap.add_argument("--batch", type=int, default=1,
                    help="parameter sets evaluated per step (BASELINE config[4] uses 64 proposals/step)")
    ap.add_argument("--condition-ms", type=float, default=60.0,
                    help="untimed evaluations for this many milliseconds before the W warmup steps of every timed "
                         "region: the card's clocks leave the idle state over the first ~15 ms of work (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-workers", type=int, default=0,
                    help="processes that sample the synthetic alignments (0 = automatic: up to 16, forked before the "
                         "first GPU call - or in-process when a profiler's tool library is injected, because rocprofv3 "
                         "--pmc initialises the GPU before Python starts and a fork from such a process can hang)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs leg (profiling runs)")
    ap.add_argument("--no-compress", action="store_true",
                    help="raw symbol stream (one step per alignment column), kernel chosen automatically")
    ap.add_argument("--mode", type=int, default=-1, help="imc_set_compression mode 0..5 (overrides --no-compress)")
    ap.add_argument("--cpu-sample-columns", type=int, default=0, help="cap on columns per CPU thread (0 = whole chunk)")
    ap.add_argument("--split-file", action="store_true",
                    help="N>1 only: ONE alignment of --columns (default 1e8) columns cut into contiguous slices, one per "
                         "rank, stitched with exact transfer operators (strong scaling; dist.SplitAlignmentLikelihood)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling of BASELINE config[3]: --chunks (default 256) chunks of --columns (default 1e7) "
                         "columns in TOTAL, dealt round-robin over the ranks; with --gpus 1 all of them on one GPU")
    ap.add_argument("--force-collective", action="store_true",
                    help="--gpus 1 only: run the step through the N-rank code path - a world-size-1 nccl (RCCL) group, the "
                         "device-output entry point on torch's stream and the all_reduce - to price that path on one GPU")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses device 0 and the reduction runs "
                         "over gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    ap.add_argument("--rehearse-cpu", action="store_true",
                    help="launcher / reduction rehearsal without any GPU: ranks evaluate a stand-in partial sum on the "
                         "host and reduce over gloo (tests/test_bench_launcher.py); prints a line marked invalid")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N rank processes ourselves.  This parent
    has not imported torch or touched HIP, so nothing GPU-related is inherited; it only forwards the exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def _gen_piece(task):
    key, n, seed = task
    from imcoalhmm_amd import synth
    d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
    return synth.sample_alignment(d[key + "_pi"], d[key + "_T"], d[key + "_E"], n, seed=seed)


def under_profiler():
    """True when a GPU profiler / tool library has been injected into this process (rocprofv3 sets these before Python
    starts, and with --pmc its library has initialised the GPU by then): no fork pool in that case."""
    env = os.environ
    if any(env.get(k) for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")):
        return True
    if any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCTRACER_")) for k in env):
        return True
    return any(t in env.get("LD_PRELOAD", "") for t in ("rocprof", "roctracer", "rocprofiler"))


def generate(requests):
    """requests: list of (tag, fixture key, columns, seed) -> {tag: uint8 array}.  Pieces of <= 1e7 columns are sampled
    in a fork pool - this runs BEFORE the first library / HIP call of the process, so the children inherit no GPU state."""
    tasks, index = [], []
    for tag, key, cols, seed in requests:
        for k, off in enumerate(range(0, cols, PIECE)):
            tasks.append((key, min(PIECE, cols - off), seed * 1000 + k))
            index.append(tag)
    workers = max(1, min(len(tasks), os.cpu_count() or 1, 16, GEN_WORKERS or 16))
    if not GEN_WORKERS and under_profiler():
        workers = 1            # the GPU is already initialised in this process: a forked child can hang
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            pieces = pool.map(_gen_piece, tasks, chunksize=1)
    else:
        pieces = [_gen_piece(t) for t in tasks]
    out = {}
    for tag, p in zip(index, pieces):
        out.setdefault(tag, []).append(p)
    return {tag: (ps[0] if len(ps) == 1 else np.concatenate(ps)) for tag, ps in out.items()}


def read_profile(lib):
    ms_p, ms_s = ctypes.c_double(), ctypes.c_double()
    n_p, n_s = ctypes.c_uint64(), ctypes.c_uint64()
    lib.imc_profile_read(ctypes.byref(ms_p), ctypes.byref(ms_s), ctypes.byref(n_p), ctypes.byref(n_s))
    return ms_p.value / max(n_p.value, 1), ms_s.value / max(n_s.value, 1)


def timed_steps(lib, step, steps, warmup, fence, agree=None):
    """W untimed + K timed calls of step(), bracketed by fence(); HIP-event kernel timing over the timed region.
    (Checked in round 3: the events cost the timed region nothing measurable - 0.2852 ms per step without them, 0.2835
    with, same process - although a rocprofv3 timeline shows ~5 us queue gaps around an event-bracketed kernel: those
    exist under the tracer only.)"""
    value = None
    # Device conditioning (untimed, before the W warmup steps; reported as config.device_conditioning_ms): the card leaves
    # its idle power state over the first ~10-15 ms of work - measured round 3 after 2 s of host-only setup: 185 -> 173 us per
    # evaluation at 20 states / 3e7 columns, 113.6 -> 107 us at 10 states / 100 x 1e6, profiles/r03_f_clock_ramp.txt - and
    # W = 5 steps of 0.3 ms end inside that ramp.  The quantity of interest is the rate an optimiser's thousands of
    # consecutive evaluations see.
    # (Several ranks: step() holds a collective, so every rank must make the same number of calls - the ranks agree after
    # each one whether anybody still needs more: `agree`.)
    t_cond = time.perf_counter()
    more = CONDITION_MS > 0
    while more:
        value = step()
        more = (time.perf_counter() - t_cond) * 1e3 < CONDITION_MS
        if agree is not None:
            more = agree(more)
    for _ in range(warmup):
        value = step()
    lib.imc_profile_enable(1)
    read_profile(lib)                                  # reset
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        value = step()
    fence()
    elapsed = time.perf_counter() - t0
    k_ms, s_ms = read_profile(lib)
    lib.imc_profile_enable(0)
    return value, elapsed, k_ms, s_ms


def main():
    args = parse_args()
    global GEN_WORKERS, CONDITION_MS
    GEN_WORKERS = args.gen_workers
    CONDITION_MS = args.condition_ms
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(world_env) if world_env is not None else 1
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with --nproc-per-node equal to --gpus" % (world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    if args.rehearse_cpu:
        return rehearse_cpu(args, rank, world)

    d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
    key = args.fixture or "iso%d_t0" % args.states
    if key + "_pi" not in d.files:
        raise SystemExit("no (pi,T,E) fixture for %d states" % args.states)
    pi, T, E = d[key + "_pi"], d[key + "_T"], d[key + "_E"]
    n_states = pi.shape[0]
    model_name = "isolation-model" if key.startswith("iso") else "initial-migration-model"
    workload_kind = args.workload if args.workload != "auto" else ("config2" if world == 1 and not args.strong else "config4-slice")
    if args.strong and (args.split_file or workload_kind == "config2"):
        raise SystemExit("--strong shards the chunks of BASELINE config[3]; it does not combine with --split-file / config2")

    from imcoalhmm_amd.dist import shard_indices, slice_bounds
    split = world > 1 and args.split_file
    if split:
        chunks_per_rank, cols = 1, args.columns or 100_000_000
        workload = "%d states, ONE %d-column synthetic alignment cut into %d contiguous slices (BASELINE config[%d] across GPUs)" % (
            n_states, cols, world, 1 if key.startswith("iso") else 2)
        seeds = [20240001]
    elif workload_kind == "config2":
        chunks_per_rank, cols = args.chunks or 1, args.columns or 100_000_000
        workload = "%s %d states, %d x %d-column synthetic pairwise alignment (BASELINE config[%d])" % (
            model_name, n_states, chunks_per_rank, cols, 1 if key.startswith("iso") else 2)
        seeds = [20240001 + (1 if not key.startswith("iso") else 0) + k for k in range(chunks_per_rank)]
        if world > 1:
            raise SystemExit("--workload config2 is a single-GPU workload (use --split-file to cut one alignment over ranks)")
    elif args.strong:
        total_chunks, cols = args.chunks or 256, args.columns or 10_000_000
        mine = shard_indices(total_chunks, rank, world)
        chunks_per_rank = len(mine)
        workload = "%s %d states, %d x %d-column synthetic chunks in total, sharded over %d GPU%s (BASELINE config[3], strong scaling)" % (
            model_name, n_states, total_chunks, cols, world, "" if world == 1 else "s")
        seeds = [20240100 + i for i in mine]
    else:
        chunks_per_rank, cols = args.chunks or 32, args.columns or 10_000_000
        workload = "%s %d states, %d x %d-column synthetic chunks sharded over %d GPU%s (BASELINE config[3] slice)" % (
            model_name, n_states, chunks_per_rank * world, cols, world, "" if world == 1 else "s")
        seeds = [20240100 + i for i in shard_indices(chunks_per_rank * world, rank, world)]
    lo, hi = slice_bounds(cols, rank, world) if split else (0, cols)

    # ---- synthetic data: everything this process will need, generated before the GPU is touched ----
    t0 = time.time()
    requests = [("main%d" % i, key, cols, sd) for i, sd in enumerate(seeds)]
    extras = world == 1 and not args.no_extra and not args.force_collective and args.batch == 1 and not args.fixture and workload_kind == "config2" \
        and not args.no_compress and args.mode < 0 and n_states == 20 and not args.columns and not args.chunks
    if extras:
        requests += [("c3", "im150_t0", 100_000_000, 20240002)]
        requests += [("c4_%d" % i, "iso20_t0", 10_000_000, 20240100 + i) for i in range(32)]
        requests += [("c5_%d" % i, "im150_t0", 1_000_000, 20240600 + i) for i in range(32)]
        requests += [("c5n20_%d" % i, "iso20_t0", 1_000_000, 20240700 + i) for i in range(32)]
        requests += [("a10_%d" % i, "iso10_t0", 1_000_000, 20240800 + i) for i in range(100)]
    data = generate(requests)
    t_gen = time.time() - t0

    import torch
    import torch.distributed as dist
    from imcoalhmm_amd import Forwarder, _capi
    from imcoalhmm_amd.dist import DistributedLikelihood, SplitAlignmentLikelihood

    dev_index = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = _capi.lib()                       # raises if the HIP library is missing (no fallback)
    _capi.check(lib.imc_set_device(dev_index))
    _capi.check(lib.imc_set_compression(args.mode if args.mode >= 0 else (0 if args.no_compress else 1)))
    backend = None
    if world == 1 and args.force_collective:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        backend = "nccl (world size 1, --force-collective)"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "gloo" if args.rehearse_on_one_gpu else "nccl"
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    t0 = time.time()
    forwarders = []
    first_chunk = None
    for i in range(len(seeds)):
        obs = data.pop("main%d" % i)
        if split:            # this rank's contiguous slice of the one alignment
            obs = obs[lo:hi]
        if first_chunk is None:
            first_chunk = obs
        forwarders.append(Forwarder.from_array(obs, 3))
    t_setup = time.time() - t0
    local_cols = sum(len(f) for f in forwarders)

    class FixedModel(object):   # the model layer stays CPU-side (north_star); the bench holds theta fixed
        def valid_parameters(self, p):
            return True

        def build_hidden_markov_model(self, p):
            return pi, T, E

    if split:
        ll = SplitAlignmentLikelihood(FixedModel(), forwarders[0], gather_device=None if args.rehearse_on_one_gpu else dev)
    else:
        ll = DistributedLikelihood(FixedModel(), forwarders, device=dev, reduce_on_host=args.rehearse_on_one_gpu,
                                   force_collective=args.force_collective)

    model_build_ms = None
    if args.batch > 1:
        pis, Ts, Es, model_build_ms = proposals(d, key, n_states, args.batch, T)

    def step():
        if args.batch > 1:
            return float(ll.forward_params_batch(pis, Ts, Es)[0])
        return ll.forward_params(pi, T, E)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: first every rank times its own shard ALONE - the same entry point (device output on torch's stream, read
    # back) without the collective, all ranks at once, each on its own GPU - which is the line's single-GPU base
    alone_elapsed = None
    if world > 1 and not split:
        def step_alone():
            if args.batch > 1:
                return float(ll.forward_params_batch(pis, Ts, Es, reduce=False)[0])
            return ll.forward_params(pi, T, E, reduce=False)
        _, alone_elapsed, _, _ = timed_steps(lib, step_alone, args.steps, args.warmup, torch.cuda.synchronize)
    def agree(flag):                                   # any rank still conditioning? (one tiny collective per call)
        rdev_ = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=rdev_)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(int(t.item()))

    value, elapsed, k_ms, s_ms = timed_steps(lib, step, args.steps, args.warmup, fence, agree if world > 1 else None)
    plan = _capi.last_plan()
    rank1_stats = None
    if world == 1 and args.batch == 1:     # one extra synchronous evaluation: the hand-off counters are read back there
        from imcoalhmm_amd.hmm import forward_chunks
        forward_chunks([f.handle for f in forwarders], pi, T, E)
        rank1_stats = dict(zip(("segments_tested", "collapsed"), _capi.last_rank1()))
    ntok0, alpha0 = forwarders[0].compressed_length(plan["token_alphabet"] or 256) if plan["vector_tokens"] else (len(forwarders[0]), 3)

    ranks = 1
    scaling_fields = {}
    if world > 1:
        ranks = dist.get_world_size()
        rdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        elapsed, total_cols, scaling_fields = scaling_report(torch, dist, rdev, elapsed, alone_elapsed, float(local_cols),
                                                             args.batch, args.steps)
    else:
        total_cols = float(local_cols)

    if rank == 0:
        cols_per_s = total_cols * args.batch * args.steps / elapsed
        k_s = k_ms * 1e-3
        alg_bytes = float(local_cols) * 1.0               # 1 B of observation stream per column (shared by a batch)
        alg_flops = float(local_cols) * args.batch * (2 * n_states * n_states + 3 * n_states)
        exe_flops = (float(plan["vector_columns"]) * (2 * n_states * n_states + 3 * n_states)
                     + float(plan["vector_tokens"]) * (2 * n_states * n_states))
        kernel_name = plan["kernels"]
        handoff = "rank1-handoff" in kernel_name
        if handoff:
            # the planned operator steps are an upper bound: segments certified rank one finish as vectors, and how
            # many do is only known at run time -> no executed-flop figure; the tails are an HBM stream instead
            exe_flops = float("nan")
        achieved_gbs = alg_bytes / k_s / 1e9 if k_s > 0 else 0.0
        traffic = lookup_traffic(kernel_name, n_states, local_cols, args.batch) if world == 1 else None
        operator_stream = None
        if "k_big_vector" in kernel_name and k_s > 0:
            # the mat-vec chain kernel is bound by streaming one NP x NP operator per chain step from L2 / Infinity Cache
            npad = 16 * ((n_states + 15) // 16)
            op_bytes = float(plan["vector_tokens"]) * npad * npad * 8.0
            operator_stream = {"bytes_per_launch": op_bytes, "memory_side_gbs": op_bytes / k_s / 1e9,
                               "pmc_fetch_gbs": (traffic / k_s / 1e9) if traffic else None,
                               "ceilings_gbs": {"l2_shared_rows": 16800.0, "infinity_cache": 8600.0, "hbm": 8000.0},
                               "note": "ceilings: MI355X_MICROARCH.md 'Indexed rows' table (chip-wide, measured)"}
        out = {
            "metric": "alignment columns/sec (forward pass), %d-state isolation HMM" % n_states,
            "value": cols_per_s,
            "unit": "columns/s",
            "n_gpus": n_gpus,
            "ranks": ranks,
            "backend": backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if (split or args.strong) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "states": n_states, "chunks_per_gpu": chunks_per_rank,
                       "columns_per_chunk": cols, "batch": args.batch, "evals_per_s": args.batch * args.steps / elapsed,
                       "segments": plan["segments"], "vectors": plan["vectors"],
                       "column_segment_len": plan["column_segment_len"], "token_segment_len": plan["token_segment_len"],
                       "compression": "pair dictionary, %d tokens" % plan["token_alphabet"] if plan["vector_tokens"] else "off",
                       "columns_per_token": (len(forwarders[0]) / max(ntok0, 1)) if plan["vector_tokens"] else 1.0,
                       "generate_s": t_gen, "setup_s": t_setup, "model_build_ms_per_hmm": model_build_ms, "loglik": value,
                       "rank1_handoff": rank1_stats, "device_conditioning_ms": CONDITION_MS},
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kernel_name, "kernel_ms": k_ms, "operator_stream": operator_stream, "stitch_ms": s_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms_covers": ("the scan launch alone (HIP events around it; its operator table is built by the "
                                     "k_z4_level2 launches in front of it - durations in profiles/)"
                                     if "k_zpropagate4" in kernel_name else "the propagate launch(es) of one evaluation"),
                "note": "north_star names the HBM roof, but with 1 B/column the path is bound by the fp64 units "
                        "(v_fma_f64 and v_mfma_f64 share them); see fp64",
                "fp64": {
                    "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                    "executed_tflops": None if handoff else (exe_flops / k_s / 1e12 if k_s > 0 else 0.0),
                    "frac_executed": None if handoff else (exe_flops / k_s / 1e12 / FP64_VALU_PEAK_TFLOPS if k_s > 0 else 0.0),
                    "algorithmic_tflops": alg_flops / k_s / 1e12 if k_s > 0 else 0.0,
                    "note": "algorithmic = (2N^2+3N) flop per column; with pair compression one executed step covers "
                            "many columns, so the algorithmic rate is not a fraction of any peak",
                },
            },
        }
        out.update(scaling_fields)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pi, T, E, first_chunk, args.cpu_sample_columns)
    if extras:
        del forwarders, ll
        ex = extra_configs(lib, d, data, fence)
        if rank == 0:
            out["extra_configs"] = ex
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.barrier()
        dist.destroy_process_group()


def scaling_report(torch, dist, rdev, elapsed, alone_elapsed, local_cols, batch, steps):
    """Collective part of an N > 1 line: (max-over-ranks elapsed, total columns, fields for the JSON line).

    `rank_ms_per_step`: every rank's own step time in the group run (a straggler shows as max >> min).
    `single_gpu_same_workload`: the ranks' shards timed alone (no collective) - value = the slowest rank's columns/s on
    its own shard; `scaling_efficiency` = group value / (sum over ranks of their alone values), i.e. what the collective
    and rank skew cost relative to N independent GPUs on this same workload (weak scaling: the driver's own curve
    should be read against THIS base, not against the N=1 line, which measures BASELINE config[1])."""
    world = dist.get_world_size()
    mine = torch.tensor([elapsed, alone_elapsed if alone_elapsed is not None else float("nan"), local_cols],
                        dtype=torch.float64, device=rdev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    rows = [p.cpu().tolist() for p in parts]
    group = [r[0] for r in rows]
    alone = [r[1] for r in rows]
    cols = [r[2] for r in rows]
    t_group = max(group)
    total_cols = float(sum(cols))
    fields = {"rank_ms_per_step": {"min": min(group) / steps * 1e3, "max": t_group / steps * 1e3,
                                   "per_rank": [g / steps * 1e3 for g in group]}}
    if alone_elapsed is not None:
        alone_rates = [c * batch * steps / a for c, a in zip(cols, alone)]
        slowest = max(range(world), key=lambda r: alone[r] / max(cols[r], 1.0))
        fields["single_gpu_same_workload"] = {
            "value": alone_rates[slowest], "unit": "columns/s", "ms_per_step": max(alone) / steps * 1e3,
            "columns": cols[slowest], "per_rank_ms_per_step": [a / steps * 1e3 for a in alone],
            "what": "every rank's own shard without the collective (same entry point: device output on torch's stream, "
                    "read back), all ranks at once, one GPU each; value = the slowest rank's rate"}
        fields["scaling_efficiency"] = (total_cols * batch * steps / t_group) / sum(alone_rates)
    return t_group, total_cols, fields


def proposals(d, key, n_states, batch, T_check=None):
    """B proposals per step (BASELINE config[4] shape): log-normal random-walk proposals around the fixture's theta
    (sd 0.1 in log space, mcmc.py:25,34-35; seed 20240500), turned into (pi, T, E) by the host-side model layer.  That
    layer stays on the CPU and outside the timed region (north_star); its cost per HMM is returned in ms."""
    from imcoalhmm_amd import models
    model = (models.IsolationModel(n_states) if key.startswith("iso")
             else models.IsolationMigrationModel(n_states // 2, n_states - n_states // 2))
    theta0 = d[key + "_theta"]
    rng = np.random.default_rng(20240500)
    thetas = theta0 * np.exp(0.1 * rng.standard_normal((batch, len(theta0))))
    thetas[0] = theta0
    model.build_batch(thetas[:2])
    tb = time.perf_counter()
    pis, Ts, Es = model.build_batch(thetas)
    ms = (time.perf_counter() - tb) * 1e3 / batch
    if T_check is not None:
        assert np.abs(Ts[0] - T_check).max() < 1e-12, "model layer disagrees with the reference fixture"
    return pis, Ts, Es, ms


def lookup_traffic(kernel_name, n_states, local_cols, batch):
    """HBM bytes per launch from committed PMC passes (rocprofv3 cannot run inside the bench): the measurement for this
    kernel / N / column count if there is one (profiles/r03_traffic_pmc.json, else an earlier round's), else None."""
    k = "%s|%d|%d" % (kernel_name, n_states, local_cols) + ("|B%d" % batch if batch > 1 else "")
    for name in ("r03_traffic_pmc.json", "r02_traffic_pmc.json", "r01_traffic_pmc.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as fh:
                rec = json.load(fh).get(k)
            if rec:
                return rec["hbm_bytes_per_launch"]
        except (OSError, ValueError):
            pass
    return None


def plain_latency_us(fn, reps):
    """Wall-clock microseconds per synchronous evaluation with NO event instrumentation: for an evaluation of a few tens
    of microseconds the two event records of the profiled region (`ms_per_step` of the same entry) are a visible share."""
    for _ in range(10):
        fn()
    blocks = []
    per = max(1, reps // 5)
    for _ in range(5):                                  # median of five blocks: one host hiccup (a 373 us reading of a
        t0 = time.perf_counter()                        # 125 us call was seen once) does not become the reported figure
        for _ in range(per):
            fn()
        blocks.append((time.perf_counter() - t0) / per * 1e6)
    return sorted(blocks)[2]


def extra_configs(lib, d, data, fence):
    """The other BASELINE configs at the size one GPU carries, a few steps each, after the headline measurement:
    config[2] (150 states, 1 x 1e8 columns), the per-GPU slice of config[3] (20 states, 32 x 1e7) and of config[4]
    (150 states, 64 proposals x 32 x 1e6).  `value` is columns/s (column-evaluations/s for the batch)."""
    from imcoalhmm_amd import Forwarder, _capi
    from imcoalhmm_amd.hmm import forward_chunks, forward_chunks_batch, recompress
    res = []

    def run(workload, fw, fn, cols, batch, steps, warmup):
        value, elapsed, k_ms, s_ms = timed_steps(lib, fn, steps, warmup, fence)
        plan = _capi.last_plan()
        res.append({"workload": workload, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
                    "value": cols * batch * steps / elapsed, "unit": "columns/s" if batch == 1 else "column-evaluations/s",
                    "kernel": plan["kernels"], "kernel_ms": k_ms, "stitch_ms": s_ms, "loglik": value,
                    "token_alphabet": plan["token_alphabet"],
                    "columns_per_token": (sum(len(f) for f in fw) / max(sum(f.compressed_length(plan["token_alphabet"])[0] for f in fw), 1))
                                         if plan["vector_tokens"] else 1.0})

    # config[0] as worded: the reference's own model size (--states defaults to 10, scripts/isolation-model.py:43) on its
    # only shipped alignment (examples/example_data.fa, pair hg18 / pantro2, 65,255 columns) - a latency, not a throughput
    pi10, T10, E10 = d["iso10_t0_pi"], d["iso10_t0_T"], d["iso10_t0_E"]
    pair = np.load(os.path.join(REPO, "tests", "golden", "example_pairs.npz"))["hg18__pantro2"]
    _capi.check(lib.imc_dictionary_reset())
    fw = [Forwarder.from_array(pair, 3)]
    h = _capi.handle_array([f.handle for f in fw])
    run("isolation-model 10 states, examples/example_data.fa hg18/pantro2, %d columns (BASELINE config[0], on the GPU)" % pair.size,
        fw, lambda: _capi.forward1(h, 1, pi10, T10, E10), float(pair.size), 1, 200, 20)
    res[-1]["us_per_evaluation"] = plain_latency_us(lambda: _capi.forward1(h, 1, pi10, T10, E10), 300)
    # what a caller of the reference's interface sees: Likelihood(model, forwarders)(theta) = host-side (pi, T, E)
    # construction (models.py; native for the small state spaces, include/imcoal_model.h) + the forward pass
    from imcoalhmm_amd import models as _models
    from imcoalhmm_amd.likelihood import Likelihood as _Likelihood
    lik = _Likelihood(_models.IsolationModel(10), fw, recompress=False)
    theta = np.array((0.001, 1000.0, 0.4))
    counter = iter(range(10 ** 9))
    res[-1]["end_to_end_us_per_likelihood_call"] = plain_latency_us(lambda: lik(theta * (1.0 + 1e-6 * next(counter))), 300)
    counter = iter(range(10 ** 9))
    res[-1]["model_build_us_per_call"] = plain_latency_us(lambda: lik.model.build_hidden_markov_model(theta * (1.0 + 1e-6 * next(counter))), 300)
    del fw, h, lik
    # the authors' own data scale: 100 files of 1 Mbp (simulations/isolation-model/simulate.sh:11), 10 states, one theta
    t0 = time.time()
    _capi.check(lib.imc_dictionary_reset())
    fw = [Forwarder.from_array(data.pop("a10_%d" % i), 3) for i in range(100)]
    recompress(fw)
    h = _capi.handle_array([f.handle for f in fw])
    run("isolation-model 10 states, 100 x 1000000-column synthetic chunks, one parameter set (the reference authors' data scale)",
        fw, lambda: _capi.forward1(h, 100, pi10, T10, E10), 1e8, 1, 20, 5)
    res[-1]["setup_s"] = time.time() - t0
    res[-1]["us_per_evaluation"] = plain_latency_us(lambda: _capi.forward1(h, 100, pi10, T10, E10), 50)
    lik = _Likelihood(_models.IsolationModel(10), fw, recompress=False)
    counter = iter(range(10 ** 9))
    res[-1]["end_to_end_us_per_likelihood_call"] = plain_latency_us(lambda: lik(theta * (1.0 + 1e-6 * next(counter))), 100)
    del lik
    # ... and a population on the same files: 64 proposals per step (PSO / GA populations are 100, MC3 rounds k chains:
    # particle_swarm.py:97-99, genetic_algorithm.py:750-754) - chunks x proposals alone are 25 rounds of workgroups here
    pis, Ts, Es, ms = proposals(d, "iso10_t0", 10, 64)
    hl = [f.handle for f in fw]
    run("isolation-model 10 states, 64 proposals/step x 100 x 1000000-column chunks (a population at the authors' data scale)",
        fw, lambda: float(forward_chunks_batch(hl, pis, Ts, Es)[0]), 1e8, 64, 5, 2)
    res[-1]["model_build_ms_per_hmm"] = ms
    del fw, h, hl
    # config[2]: ~150 states (IsolationMigrationModel(75, 75)), one 1e8-column alignment
    pi, T, E = d["im150_t0_pi"], d["im150_t0_T"], d["im150_t0_E"]
    t0 = time.time()
    _capi.check(lib.imc_dictionary_reset())       # every configuration compresses with a dictionary of its own data
    fw = [Forwarder.from_array(data.pop("c3"), 3)]
    h = [f.handle for f in fw]
    run("initial-migration-model 150 states, 1 x 100000000-column synthetic alignment (BASELINE config[2])", fw,
        lambda: forward_chunks(h, pi, T, E), 1e8, 1, 3, 2)
    res[-1]["setup_s"] = time.time() - t0
    res[-1]["rank1_handoff"] = dict(zip(("segments_tested", "collapsed"), _capi.last_rank1()))
    del fw, h
    # config[3] slice: 20 states, 32 x 1e7 columns (what one of 8 GPUs holds of the 256 chunks)
    pi20, T20, E20 = d["iso20_t0_pi"], d["iso20_t0_T"], d["iso20_t0_E"]
    t0 = time.time()
    _capi.check(lib.imc_dictionary_reset())
    fw = [Forwarder.from_array(data.pop("c4_%d" % i), 3) for i in range(32)]
    recompress(fw)                                # what Likelihood(model, forwarders) does: one dictionary from all chunks
    h = [f.handle for f in fw]
    run("isolation-model 20 states, 32 x 10000000-column synthetic chunks (per-GPU slice of BASELINE config[3])", fw,
        lambda: forward_chunks(h, pi20, T20, E20), 3.2e8, 1, 5, 2)
    res[-1]["setup_s"] = time.time() - t0
    del fw, h
    # config[4] slice: 150 states, 64 proposals per step x 32 x 1e6 columns
    pis, Ts, Es, ms = proposals(d, "im150_t0", 150, 64)
    t0 = time.time()
    _capi.check(lib.imc_dictionary_reset())
    fw = [Forwarder.from_array(data.pop("c5_%d" % i), 3) for i in range(32)]
    recompress(fw)
    h = [f.handle for f in fw]
    run("initial-migration-model 150 states, 64 proposals/step x 32 x 1000000-column chunks (per-GPU slice of BASELINE config[4])",
        fw, lambda: float(forward_chunks_batch(h, pis, Ts, Es)[0]), 3.2e7, 64, 2, 1)
    res[-1]["setup_s"] = time.time() - t0
    res[-1]["model_build_ms_per_hmm"] = ms
    del fw, h
    # config[4] as BASELINE.json literally words it for the reference's model sizes: 20 states, same batch shape
    pis, Ts, Es, ms = proposals(d, "iso20_t0", 20, 64)
    t0 = time.time()
    _capi.check(lib.imc_dictionary_reset())
    fw = [Forwarder.from_array(data.pop("c5n20_%d" % i), 3) for i in range(32)]
    recompress(fw)
    h = [f.handle for f in fw]
    run("isolation-model 20 states, 64 proposals/step x 32 x 1000000-column chunks (BASELINE config[4] shape at 20 states)",
        fw, lambda: float(forward_chunks_batch(h, pis, Ts, Es)[0]), 3.2e7, 64, 5, 2)
    res[-1]["setup_s"] = time.time() - t0
    res[-1]["model_build_ms_per_hmm"] = ms
    del fw, h
    return res


def rehearse_cpu(args, rank, world):
    """Launcher / sharding / reduction rehearsal with no GPU anywhere (tests): every rank owns its config-4 shard of
    chunk ids, its "partial log-likelihood" is a stand-in computed on the host, the reduction is the real
    DistributedLikelihood path over gloo.  The printed line is marked invalid and carries no throughput claim."""
    import torch
    import torch.distributed as dist
    from imcoalhmm_amd.dist import DistributedLikelihood, shard_indices
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    if args.strong:
        total_chunks = args.chunks or 8
        mine = shard_indices(total_chunks, rank, world)
    else:
        total_chunks = (args.chunks or 4) * world
        mine = shard_indices(total_chunks, rank, world)

    class FixedModel(object):
        def valid_parameters(self, p):
            return True

        def build_hidden_markov_model(self, p):
            return np.ones(2) / 2, np.eye(2), np.ones((2, 3)) / 3

    def local_eval(pis, Ts, Es):
        return torch.tensor([-float(sum(1000 + i for i in mine))] * pis.shape[0], dtype=torch.float64)

    ll = DistributedLikelihood(FixedModel(), [], local_eval=local_eval)
    pi, T, E = FixedModel().build_hidden_markov_model(None)
    steps = max(args.steps, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        alone_value = ll.forward_params(pi, T, E, reduce=False)
    alone_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        value = ll(np.array([1.0]))
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    want = -float(sum(1000 + i for i in range(total_chunks)))
    ranks = dist.get_world_size() if world > 1 else 1
    fields = {}
    if world > 1:
        elapsed, _, fields = scaling_report(torch, dist, torch.device("cpu"), elapsed, alone_elapsed, float(len(mine)), 1, steps)
    if rank == 0:
        rec = {"metric": "launcher rehearsal (no GPU, INVALID as a measurement)", "value": 0.0, "unit": "columns/s",
               "n_gpus": args.gpus, "ranks": ranks, "backend": "gloo" if world > 1 else None,
               "steps": args.steps, "warmup": args.warmup, "valid": False,
               "scaling": "strong" if args.strong else "weak",
               "config": {"workload": "rehearsal", "chunks": total_chunks, "chunks_this_rank": len(mine), "loglik": value,
                          "loglik_expected": want, "alone_partial": alone_value}}
        rec.update(fields)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if value == want else 1


def cpu_baseline(pi, T, E, obs, cols_per_thread):
    """The CPU oracle (a port: ziphmm itself is not installable offline) timed on this host's cores on the
    SAME alignment: it is cut into one slice per thread (each slice evaluated as its own chunk - timing
    only) and the evaluation is repeated until a few seconds of wall time have accumulated."""
    from oracle import oracle_lib
    oracle_lib.build()
    cores = min(os.cpu_count() or 1, oracle_lib.max_threads(), 64)
    n = obs.size // cores if cores else obs.size
    if cols_per_thread:
        n = min(n, cols_per_thread)
    slices = [obs[k * n:(k + 1) * n] for k in range(cores)]
    zips = [oracle_lib.Zip(s, 3) for s in slices]                 # one-time preprocessing, not timed (hmm.py:16)

    def timed(fn, budget_s):
        fn()                                                       # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or reps >= 500:
                return el / reps

    t_zip = timed(lambda: oracle_lib.forward_chunks_mt(pi, T, E, slices, threads=cores, zips=zips), 6.0)
    t_plain = timed(lambda: oracle_lib.forward_chunks_mt(pi, T, E, slices, threads=cores), 4.0)
    total = float(n * cores)
    best = min(t_zip, t_plain)
    # ... and the SAME computation as the GPU's: the whole alignment as one chain, one core (the like-for-like figure)
    whole = obs if not cols_per_thread else obs[:cols_per_thread * cores]
    zw = oracle_lib.Zip(whole, 3)
    t_one = timed(lambda: oracle_lib.forward_chunks_mt(pi, T, E, [whole], threads=1, zips=[zw]), 3.0)
    return {"value": total / best, "unit": "columns/s", "cores": cores, "kind": "port",
            "single_chain": {"value": float(whole.size) / t_one, "unit": "columns/s", "cores": 1,
                             "sample": "the whole %d-column alignment as ONE chain on one core (what the GPU computes), "
                                       "zipHMM-style compressed forward, compression ratio %.1fx" % (whole.size, whole.size / max(zw.length, 1))},
            "sample": "%d threads x %d columns of the same synthetic alignment (%.0f%% of it), each slice its own chunk "
                      "(so NOT the same computation as the GPU's single chain - a stated baseline, not a like-for-like "
                      "ratio), repeated for ~10 s; zipHMM-style compressed forward %.3g col/s (compression ratio %.1fx), "
                      "textbook scaled forward %.3g col/s; CPU restatement of the ziphmm forward, ziphmm itself is not "
                      "installable offline" % (cores, n, 100.0 * n * cores / obs.size, total / t_zip,
                                               n / max(zips[0].length, 1), total / t_plain)}


if __name__ == "__main__":
    sys.exit(main())
