# the round's closing measurement: full GPU suite, the driver's bench invocation, kernel traces, per-layer gradient benches
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
(timeout -k 5 420 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/r5_final_pytest.log
export BTX_TRACE_STATS_ONLY=1
timeout -k 5 900 python tools/refresh_profiles.py bench train wgrad trace > $O/r5_final_refresh.log 2>&1
tail -3 $O/r5_final_pytest.log; tail -8 $O/r5_final_refresh.log; python - <<'PY'
import json
d = json.load(open("gpurun_out/profiles/r05_bench_line.json"))
print(d["value"], d["roofline"]["frac"], d["ms_per_step"], {k: v.get("value", v.get("ms_per_step")) for k, v in d.get("extra", {}).items()})
PY
