# closing measurement after the TR instantiation of contract_dma_kernel: full GPU suite, the driver's bench invocation, its kernel trace
O=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
(timeout -k 5 420 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/r5_final_pytest.log
export BTX_TRACE_STATS_ONLY=1
timeout -k 5 700 python tools/refresh_profiles.py bench trace > $O/r5_final_refresh.log 2>&1
tail -3 $O/r5_final_pytest.log | head -2; tail -3 $O/r5_final_refresh.log; python - <<'PY'
import json
d = json.load(open("gpurun_out/profiles/r05_bench_line.json"))
print(d["value"], d["roofline"]["frac"], d["ms_per_step"], {k: v.get("value", v.get("ms_per_step")) for k, v in d.get("extra", {}).items()})
PY
grep "k1x1" gpurun_out/profiles/r05_bench_summary.txt | cut -c1-120
