cd $GRAFT_REPO_ROOT; O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "inear or mlp or lanes or presample or fc or smoke or mc or lstm or golden or batched" > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -6
python tools/r05_experiments/e8.py 2>&1 | tail -1
python bench.py --steps 20 --warmup 20 --no-extras --no-cpu-baseline --no-traffic --no-sustain 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['rows'])"
