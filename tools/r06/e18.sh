# E18 (round 6): the fc launch (Linear 512 -> 1000, 64 rows per MC sample lane) under the throughput plan's K split: split until a lane
# has >= BTX_TP_MINWG workgroups (shipped: 64 -> 4 splits of 4 stages + a reduce launch)
# bash tools/build_variants.sh tune "-DBTX_TUNING"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e18; mkdir -p $O
cd $R
for i in 1 2; do
for W in 64 32 16; do
  BTX_TP_MINWG=$W BTX_LIB=build_variants/libbtx_tune.so python bench.py --no-extras --no-cpu-baseline --no-traffic --no-sustain 2>/dev/null | tail -1 > /dev/null
  python - $W >> $O/fc.txt <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_detail.json"))
r = d["roofline"]
fc = [x["us"] for x in r["per_launch"] if "cout1000" in x["launch"]]
print("MINWG", sys.argv[1], "value", d["value"], "fc us", fc, "kernel_us_per_step", r["kernel_us_per_step"])
PY
done; done
cat $O/fc.txt
