"""bench.py must never label a one-GPU measurement as N GPUs: run without a torchrun environment, `--gpus N` starts
its own N rank processes (before anything touches the GPU); under a launcher, WORLD_SIZE must equal --gpus.  The
no-GPU rehearsal mode exercises launcher, static sharding and the gloo reduction on CPU (world 2)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_self_launch_world2_gloo():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-cpu", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1                                   # rank 0 only
    rec = json.loads(line[0])
    assert rec["n_gpus"] == 2 and rec["ranks"] == 2 and rec["backend"] == "gloo"
    assert rec["config"]["loglik"] == rec["config"]["loglik_expected"]


def test_world_size_mismatch_is_refused():
    env = _clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--rehearse-cpu"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_single_rank_reports_one_rank():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--rehearse-cpu"], capture_output=True, text=True,
                         timeout=120, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["ranks"] == 1 and rec["backend"] is None


def test_multi_rank_line_carries_its_own_single_gpu_base():
    """An N > 1 line must be judgeable by itself: the ranks' shards timed alone (no collective), the efficiency against
    that base, and every rank's step time; --strong deals a FIXED number of chunks over the ranks."""
    for extra, total, scaling in (([], 8, "weak"), (["--strong", "--chunks", "6"], 6, "strong")):
        out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rehearse-cpu", "--steps", "3", "--warmup", "0"] + extra,
                             capture_output=True, text=True, timeout=300, env=_clean_env())
        assert out.returncode == 0, out.stderr[-2000:]
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
        assert rec["scaling"] == scaling and rec["config"]["chunks"] == total
        assert rec["config"]["chunks_this_rank"] == total // 2
        assert rec["config"]["loglik"] == rec["config"]["loglik_expected"]
        base = rec["single_gpu_same_workload"]
        assert base["value"] > 0 and base["ms_per_step"] > 0 and len(base["per_rank_ms_per_step"]) == 2
        assert 0 < rec["scaling_efficiency"] < 10
        per = rec["rank_ms_per_step"]
        assert len(per["per_rank"]) == 2 and per["min"] <= per["max"] and per["max"] == max(per["per_rank"])
        # rank 0's partial alone is only its own shard: chunks 0, 2, 4, ...
        assert rec["config"]["alone_partial"] == -float(sum(1000 + i for i in range(0, total, 2)))
