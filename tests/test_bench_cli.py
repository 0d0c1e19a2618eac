"""bench.py command line: `--gpus N` launches N ranks itself and can never print a line for fewer ranks than asked for
(CPU / gloo dry run of the multi-rank protocol; the GPU path is the same code with backend nccl = RCCL)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=e, capture_output=True,
                          text=True, timeout=600)


def test_bench_self_launches_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dry_run"] is True and d["steps"] == 3
    assert d["value"] > 0 and abs(d["value"] - 6 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]


def test_bench_refuses_a_rank_count_mismatch():
    # a single process claiming --gpus 2 inside a 1-rank "group" must abort without a JSON line
    r = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "{" not in r.stdout
    # no GPU here: the real (non dry-run) multi-GPU launch refuses before spawning anything
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "{" not in r.stdout and "GPU" in r.stderr


def test_bench_dry_run_strong_scaling_ragged_shard():
    """--scaling strong: 5 MC samples over 2 ranks (3 + 2: a rank with fewer samples), one all-reduce; the merged statistics
    equal the single-process evaluation of the same 5 samples (checked inside the dry run, reported on the line)"""
    r = _run(["--gpus", "2", "--dry-run", "--scaling", "strong", "--total-samples", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong" and d["total_samples"] == 5
    assert d["steps"] == 3 and d["merged_equals_single_process"] is True
    assert abs(d["value"] - 5 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    # more ranks than samples: a rank with nothing to do still joins the collective
    r = _run(["--gpus", "2", "--dry-run", "--scaling", "strong", "--total-samples", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["total_samples"] == 1 and d["merged_equals_single_process"] is True
