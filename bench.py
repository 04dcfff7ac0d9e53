#!/usr/bin/env python3
"""bench.py - alignment columns/s of the MI355X forward log-likelihood path (BASELINE.json metric).

A "step" is one evaluation of the summed log-likelihood (one pass of the hot path) over this
job's synthetic alignment chunks, which are resident in HBM before the timed region starts; the
per-step host input is only (pi, T, E) (a few KB), exactly what Forwarder.forward receives.

  N=1  : BASELINE config[1] - isolation model, 20 states, 1 x 100 Mbp synthetic pairwise alignment
  N>1  : BASELINE config[3] sliced per GPU - 32 x 10 Mbp chunks per rank (256 chunks at N=8),
         chunks sharded statically, one RCCL all-reduce(sum) of the partial log-likelihoods per step.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` and, at N=1,
`cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 fp64 FMA lanes x 2 x 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--states", type=int, default=20)
    ap.add_argument("--fixture", type=str, default="",
                    help="(pi,T,E) fixture key in tests/golden/hmm_params.npz, e.g. im150_t0 (default iso<states>_t0)")
    ap.add_argument("--columns", type=int, default=0, help="override columns per chunk")
    ap.add_argument("--chunks", type=int, default=0, help="override chunks per rank")
    ap.add_argument("--batch", type=int, default=1,
                    help="parameter sets evaluated per step (BASELINE config[4] uses 64 proposals/step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-compress", action="store_true",
                    help="raw symbol stream (one step per alignment column), kernel chosen automatically")
    ap.add_argument("--mode", type=int, default=-1, help="imc_set_compression mode 0..5 (overrides --no-compress)")
    ap.add_argument("--cpu-sample-columns", type=int, default=0, help="cap on columns per CPU thread (0 = whole chunk)")
    ap.add_argument("--split-file", action="store_true",
                    help="N>1 only: ONE alignment of --columns (default 1e8) columns cut into contiguous slices, one per "
                         "rank, stitched with exact transfer operators (strong scaling; dist.SplitAlignmentLikelihood)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a 1-GPU box: every rank uses device 0 and the reduction runs "
                         "over gloo (RCCL refuses two ranks on one device); never used for reported numbers")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if world != n_gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, n_gpus))

    import torch
    import torch.distributed as dist
    from imcoalhmm_amd import Forwarder, _capi, synth
    from imcoalhmm_amd.dist import DistributedLikelihood, SplitAlignmentLikelihood, shard_indices, slice_bounds

    dev_index = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    lib = _capi.lib()                       # raises if the HIP library is missing (no fallback)
    _capi.check(lib.imc_set_device(dev_index))
    _capi.check(lib.imc_set_compression(args.mode if args.mode >= 0 else (0 if args.no_compress else 1)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    # ---- workload -------------------------------------------------------------------------------
    d = np.load(os.path.join(REPO, "tests", "golden", "hmm_params.npz"))
    key = args.fixture or "iso%d_t0" % args.states
    if key + "_pi" not in d.files:
        raise SystemExit("no (pi,T,E) fixture for %d states" % args.states)
    pi, T, E = d[key + "_pi"], d[key + "_T"], d[key + "_E"]
    n_states = pi.shape[0]
    if world == 1:
        chunks_per_rank = args.chunks or 1
        cols = args.columns or 100_000_000
        workload = "%s %d states, %d x %d-column synthetic pairwise alignment (BASELINE config[%d])" % (
            "isolation-model" if key.startswith("iso") else "initial-migration-model", n_states, chunks_per_rank, cols,
            1 if key.startswith("iso") else 2)
        seeds = [20240001 + k for k in range(chunks_per_rank)]
    elif args.split_file:
        chunks_per_rank = 1
        cols = args.columns or 100_000_000
        workload = "%d states, ONE %d-column synthetic alignment cut into %d contiguous slices (BASELINE config[%d] across GPUs)" % (
            n_states, cols, world, 1 if key.startswith("iso") else 2)
        seeds = [20240001]
    else:
        chunks_per_rank = args.chunks or 32
        cols = args.columns or 10_000_000
        workload = "isolation-model %d states, %d x %d-column synthetic chunks sharded over %d GPUs (BASELINE config[3] slice)" % (
            n_states, chunks_per_rank * world, cols, world)
        seeds = [20240100 + i for i in shard_indices(chunks_per_rank * world, rank, world)]
    split = world > 1 and args.split_file
    lo, hi = slice_bounds(cols, rank, world) if split else (0, cols)

    t0 = time.time()
    forwarders = []
    first_chunk = None
    for sd in seeds:
        # generated in 1e7-column pieces to bound host memory
        parts = [synth.sample_alignment(pi, T, E, min(10_000_000, cols - off), seed=sd * 1000 + k)
                 for k, off in enumerate(range(0, cols, 10_000_000)) if off < hi and off + 10_000_000 > lo]
        obs = parts[0] if len(parts) == 1 else np.concatenate(parts)
        if split:            # this rank's contiguous slice of the one alignment
            first_piece = (lo // 10_000_000) * 10_000_000
            obs = obs[lo - first_piece:hi - first_piece]
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
        ll = DistributedLikelihood(FixedModel(), forwarders, device=dev, reduce_on_host=args.rehearse_on_one_gpu)

    model_build_ms = None
    if args.batch > 1:
        # B proposals per step (BASELINE config[4] shape): log-normal random-walk proposals around the
        # fixture's theta (sd 0.1 in log space, mcmc.py:25,34-35; seed 20240500), turned into (pi, T, E) by
        # the host-side model layer.  That layer stays on the CPU and outside the timed region (north_star);
        # its cost per HMM is reported in config.model_build_ms_per_hmm.
        from imcoalhmm_amd import models
        model = (models.IsolationModel(n_states) if key.startswith("iso")
                 else models.IsolationMigrationModel(n_states // 2, n_states - n_states // 2))
        theta0 = d[key + "_theta"]
        rng = np.random.default_rng(20240500)
        thetas = theta0 * np.exp(0.1 * rng.standard_normal((args.batch, len(theta0))))
        thetas[0] = theta0
        model.build_batch(thetas[:2])
        tb = time.perf_counter()
        pis, Ts, Es = model.build_batch(thetas)
        model_build_ms = (time.perf_counter() - tb) * 1e3 / args.batch
        assert np.abs(Ts[0] - T).max() < 1e-12, "model layer disagrees with the reference fixture"

    def step():
        if args.batch > 1:
            return float(ll.forward_params_batch(pis, Ts, Es)[0])
        return ll.forward_params(pi, T, E)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        value = step()
    lib.imc_profile_enable(1)
    ms_p, ms_s = ctypes.c_double(), ctypes.c_double()
    n_p, n_s = ctypes.c_uint64(), ctypes.c_uint64()
    lib.imc_profile_read(ctypes.byref(ms_p), ctypes.byref(ms_s), ctypes.byref(n_p), ctypes.byref(n_s))  # reset
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        value = step()
    fence()
    elapsed = time.perf_counter() - t0
    lib.imc_profile_read(ctypes.byref(ms_p), ctypes.byref(ms_s), ctypes.byref(n_p), ctypes.byref(n_s))
    lib.imc_profile_enable(0)
    plan = _capi.last_plan()
    rank1_stats = None
    if world == 1 and args.batch == 1:     # one extra synchronous evaluation: the hand-off counters are read back there
        from imcoalhmm_amd.hmm import forward_chunks
        forward_chunks([f.handle for f in forwarders], pi, T, E)
        rank1_stats = dict(zip(("segments_tested", "collapsed"), _capi.last_rank1()))
    ntok0, alpha0 = forwarders[0].compressed_length(plan["token_alphabet"] or 256) if plan["vector_tokens"] else (len(forwarders[0]), 3)

    if world > 1:
        rdev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([float(local_cols)], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_cols = float(tot.item())
    else:
        total_cols = float(local_cols)

    if rank == 0:
        cols_per_s = total_cols * args.batch * args.steps / elapsed
        k_ms = ms_p.value / max(n_p.value, 1)            # average propagate-kernel duration (HIP events)
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
        # HBM traffic per launch comes from PMC passes (rocprofv3 cannot run inside the bench): the committed
        # measurement for this kernel / N / column count, if any (profiles/r01_traffic_pmc.json), else null
        traffic = None
        try:
            with open(os.path.join(REPO, "profiles", "r01_traffic_pmc.json")) as fh:
                rec = json.load(fh).get("%s|%d|%d" % (kernel_name, n_states, local_cols)
                                        + ("|B%d" % args.batch if args.batch > 1 else ""))
            if rec and world == 1:
                traffic = rec["hbm_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        operator_stream = None
        if "k_big_vector" in kernel_name and k_s > 0:
            # the mat-vec chain kernel is bound by streaming one NP x NP operator per chain step from L2 / Infinity Cache
            npad = 16 * ((n_states + 15) // 16)
            op_bytes = float(plan["vector_tokens"]) * npad * npad * 8.0
            operator_stream = {"bytes_per_launch": op_bytes, "achieved_gbs": op_bytes / k_s / 1e9,
                               "hbm_side_gbs": (traffic / k_s / 1e9) if traffic else None,
                               "ceilings_gbs": {"l2_shared_rows": 16800.0, "infinity_cache": 8600.0, "hbm": 8000.0},
                               "note": "ceilings: MI355X_MICROARCH.md 'Indexed rows' table (chip-wide, measured)"}
        out = {
            "metric": "alignment columns/sec (forward pass), %d-state isolation HMM" % n_states,
            "value": cols_per_s,
            "unit": "columns/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "states": n_states, "chunks_per_gpu": chunks_per_rank,
                       "columns_per_chunk": cols, "batch": args.batch, "evals_per_s": args.batch * args.steps / elapsed,
                       "segments": plan["segments"], "vectors": plan["vectors"],
                       "column_segment_len": plan["column_segment_len"], "token_segment_len": plan["token_segment_len"],
                       "compression": "pair dictionary, %d tokens" % plan["token_alphabet"] if plan["vector_tokens"] else "off",
                       "columns_per_token": (len(forwarders[0]) / max(ntok0, 1)) if plan["vector_tokens"] else 1.0,
                       "setup_s": t_setup, "model_build_ms_per_hmm": model_build_ms, "loglik": value,
                       "rank1_handoff": rank1_stats},
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kernel_name, "kernel_ms": k_ms, "operator_stream": operator_stream, "stitch_ms": ms_s.value / max(n_s.value, 1),
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "north_star names the HBM roof, but with 1 B/column the path is fp64-VALU/latency bound; "
                        "see fp64_valu",
                "fp64_valu": {
                    "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                    "algorithmic_tflops": alg_flops / k_s / 1e12 if k_s > 0 else 0.0,
                    "executed_tflops": None if handoff else (exe_flops / k_s / 1e12 if k_s > 0 else 0.0),
                    "frac_algorithmic": alg_flops / k_s / 1e12 / FP64_VALU_PEAK_TFLOPS if k_s > 0 else 0.0,
                    "frac_executed": None if handoff else (exe_flops / k_s / 1e12 / FP64_VALU_PEAK_TFLOPS if k_s > 0 else 0.0),
                },
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pi, T, E, first_chunk, args.cpu_sample_columns)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    return {"value": total / best, "unit": "columns/s", "cores": cores, "kind": "port",
            "sample": "%d threads x %d columns of the same synthetic alignment (%.0f%% of it), each slice its own chunk, "
                      "repeated for ~10 s; zipHMM-style compressed forward %.3g col/s (compression ratio %.1fx), textbook "
                      "scaled forward %.3g col/s; CPU restatement of the ziphmm forward, ziphmm itself is not installable "
                      "offline" % (cores, n, 100.0 * n * cores / obs.size, total / t_zip, n / max(zips[0].length, 1),
                                   total / t_plain)}


if __name__ == "__main__":
    main()
