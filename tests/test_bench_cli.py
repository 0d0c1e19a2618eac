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


def _stub_result():
    """a full bench result of the size a real run produces (per-launch tables, traffic breakdown, prose)"""
    rows = [{"launch": "flipout k3x3 s1 cin%d cout%d M4014080" % (c, c), "lanes": 20, "weights": 9 * c * c, "us": 580.123456,
             "gflop": 591.9, "mbytes": 100.0, "tflops": 1019.123456, "tbs": 1.0, "bound": "mfma", "frac": 0.40786170633,
             "sampling_us": 12.0, "frac_incl_sampling": 0.3996028124934605} for c in (64, 64, 64, 64, 128, 128, 128, 256, 256, 256, 512, 512, 512)]
    rows += [{"launch": "flipout k7x7 s2 cin4 cout64 M16056320", "lanes": 20, "weights": 9408, "us": 1200.0, "gflop": 302.0,
              "mbytes": 400.0, "tflops": 250.0, "tbs": 0.3, "bound": "mfma", "frac": 0.204, "sampling_us": 0.1,
              "frac_incl_sampling": 0.204}] * 8
    layer = {"launch": "flipout 3x3 s1 cin512 cout512 7x7, batch 64 x 20 lanes", "hbm_bytes": 617211136.0,
             "by_kernel": {"sampling": 146396160.0, "contraction": 470814976.0}, "algorithmic_bytes": 147324928,
             "ratio": 4.189454862655694}
    sub = {"workload": "w" * 200, "ms_per_step": 2.9655160615220666, "value": 337.20943648733595, "unit": "MC-samples/s",
           "dtype": "bf16", "kl": 207.42, "lanes": 16, "lane_mode": "launch", "ms_per_step_runs": [2.96516] * 5,
           "kl_rel_err": 7.356456047897003e-08, "logits_rel_l2_vs_unfused_f32": 0.0038692899979650974,
           "gflop_per_step": 2093.66, "achieved_e2e_tflops": 706.0, "frac_e2e": 0.28240107902129064,
           "dominant_kernel_tflops": 1074.27, "dominant_kernel_frac": 0.4297107831301862, "kernel_us_per_step": 2884.9,
           "per_launch": [dict(r, count=3, share_of_kernel_time=0.05) for r in rows]}
    return {
        "metric": "MC-samples/sec (Bayesian-ResNet18, 224^2, bs=64)", "value": 1885.098527652225, "unit": "MC-samples/s",
        "n_gpus": 1, "rccl_ranks": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.5304762511514127,
        "ms_per_step_runs": [0.5304762511514127] * 5, "timed_regions": 5, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "dnn_to_bnn(resnet18) Flipout, 224x224, batch 64, " + "x" * 250, "global_batch": 64,
                   "parallelism": "mc-sample-shard x1"},
        "image_samples_per_s": 120646.3, "kl": 55.67, "kl_rel_err": 6.851739457947929e-08,
        "sustained": {"value": 1800.123456, "seconds": 2.01, "mc_samples": 3600, "shader_clock_ghz": 1.65432, "package_w": 1390.1234,
                      "source": "hwmon", "samples": 200},
        "roofline": {"bound": "mfma", "achieved": 999.0070312336513, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.3996028124934605,
                     "traffic": 617211136.0, "traffic_detail": dict(layer, layers={"layer4": layer, "layer3": layer, "layer1": layer},
                                                                   how="h" * 300), "traffic_source": "measured in this run",
                     "kernel": "k" * 300, "kernel_name": "contract_taps_kernel<bf16,Flipout,3,3>",
                     "achieved_contraction_only": 1019.65, "frac_contraction_only": 0.40786170633097557,
                     "mc_samples_per_launch": 20, "avg_launch_us": 580.49, "avg_sampling_share_us": 11.99,
                     "launches_per_forward": 21, "achieved_e2e": 875.44, "frac_e2e": 0.3501779957583409,
                     "shader_clock_ghz": 1.65432, "package_w": 1390.1234, "measured": "m" * 600, "per_launch": rows,
                     "rows": {"k3x3_s1_%d" % h: 0.4 for h in (56, 28, 14, 7)}, "frac_min_dominant_row": 0.341,
                     "modes_within_1e-4": {"f32": {"dominant_kernel_frac": 0.6285, "dominant_kernel_tflops": 98.87, "peak": 157.3,
                                                   "value": 193.28, "logits_rel_l2": 1.1e-6, "frac_e2e": 0.57},
                                           "bf16x3": {"dominant_kernel_frac": 0.4464, "dominant_kernel_tflops": 372.0,
                                                      "peak": 833.33, "value": 693.07, "logits_rel_l2": 6.27e-06, "frac_e2e": 0.386}}},
        "logits_rel_l2_vs_unfused_f32": 0.0038291679229587317,
        "extra": {k: dict(sub) for k in ("cfg3", "cfg4_f32_parity_mode", "cfg4_bf16x3", "cfg2", "cfg5", "cfg5_bf16x3",
                                         "cfg4_strong_shape_4_per_rank", "train_step")},
        "cpu_baseline": {"value": 0.5261874214880735, "unit": "MC-samples/s", "cores": 32, "kind": "port",
                         "cpu": "AMD EPYC 9575F 64-Core Processor", "host_threads": 256, "runs": [{"cores": 256}, {"cores": 32}],
                         "sample": "4 timed MC forwards (+1 warm-up) per thread count of ResNet18-Flipout bs64 224^2 f32, " + "s" * 120},
        "gpu_over_cpu": 3582.56}


def test_bench_final_line_is_compact_and_last(tmp_path, capsys, monkeypatch):
    """the driver parses ONE stdout line and keeps an 8 KB tail: the final line must stay <= 6 KB whatever the tables hold,
    be the LAST stdout line, and carry `roofline` and `cpu_baseline` (round 4's 25 KB line came back `parsed: null`)"""
    sys.path.insert(0, ROOT)
    import bench
    full = _stub_result()
    assert len(json.dumps(full)) > 20000  # the stub is as big as a real result
    monkeypatch.setattr(bench, "DETAIL_PATH", str(tmp_path / "sub" / "bench_detail.json"))
    bench.emit(full)
    cap = capsys.readouterr()
    out_lines = cap.out.splitlines()
    assert len(out_lines) == 1, "stdout must hold the one JSON line only"
    line = out_lines[-1]
    assert len(line.encode()) <= bench.LINE_LIMIT == 6144
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio", "algorithmic_bytes",
              "frac_contraction_only", "avg_launch_us", "mc_samples_per_launch", "shader_clock_ghz", "package_w",
              "modes_within_1e-4"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert set(rf["modes_within_1e-4"]) == {"f32", "bf16x3"}
    for m in rf["modes_within_1e-4"].values():
        assert {"value", "logits_rel_l2", "dominant_kernel_frac"} <= set(m)
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb)
    assert "workload" in d["config"] and "model" not in d["config"]
    assert '"per_launch"' not in json.dumps(d)
    # the tables are not lost: the detail file holds the full result, stderr echoes it
    det = json.load(open(tmp_path / "sub" / "bench_detail.json"))
    assert len(det["roofline"]["per_launch"]) == 21 and '"per_launch"' in cap.err
    # a pathological result (huge strings in kept fields) still yields a parseable line within the limit
    full["extra"]["cfg5"]["value"] = 1.0
    full["config"]["workload"] = "y" * 3000
    full["cpu_baseline"]["sample"] = "z" * 2500
    s = bench.compact_line(full)
    assert len(s) <= bench.LINE_LIMIT and "roofline" in json.loads(s) and "cpu_baseline" in json.loads(s)


def test_timed_regions_settle_and_every_region_is_exactly_the_steps():
    """bench.timed_mc: each timed region runs EXACTLY the given MC sample indices between barrier + synchronize; with
    min_seconds the same region repeats back to back until the regions add up to that long; bench.settled() = the median of
    the second half (the first half carries the clock ramp)"""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeRunner:
        def __init__(self):
            self.packed = torch.zeros(4)
            self.calls = []

        def run(self, idx):
            self.calls.append(list(idx))
            import time
            time.sleep(0.002)

        def zero(self):
            self.packed.zero_()

        def fold(self):
            pass

    r = FakeRunner()
    runs = bench.timed_mc(r, [5, 6, 7], [100], 1, torch.device("cpu"), repeats=3, min_seconds=0.05)
    assert r.calls[0] == [100] and all(c == [5, 6, 7] for c in r.calls[1:])
    assert len(runs) == len(r.calls) - 1 and len(runs) >= 3 and 0.04 < sum(runs) < 0.5
    r2 = FakeRunner()
    assert len(bench.timed_mc(r2, [1], [0], 1, torch.device("cpu"), repeats=4)) == 4   # no settling asked for: `repeats` regions
    assert bench.settled([9.0, 8.0, 7.0, 1.0, 2.0, 3.0, 2.5, 2.2]) == 2.5              # median of the second half [2.0, 3.0, 2.5, 2.2] -> sorted[2]
    assert bench.settled([3.0, 1.0, 2.0]) == 2.0                                      # fewer than 8 regions: the median of all
