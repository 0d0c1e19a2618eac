R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $R/tools/train_profile.py > $O/train.log 2>&1
db=$(find $O/prof -name "*.db" | head -1); python $R/tools/trace_report.py $db > $O/train_trace.txt
rm -rf $O/prof
head -45 $O/train_trace.txt | cut -c1-150
tail -3 $O/train.log | cut -c1-400
cd $R
python - <<'PY'
import torch, time, sys
sys.path.insert(0,'.')
import bench
# mc_accumulate timing at 20 lanes
from bayesian_torch_amd import mc
dev=torch.device('cuda:0')
lg=torch.randn(20*64,1000,device=dev).to(torch.bfloat16)
pk=torch.zeros(mc.packed_numel(64,1000),device=dev)
for _ in range(3): mc.accumulate_lanes(pk,lg,20,0.5)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): mc.accumulate_lanes(pk,lg,20,0.5)
e1.record(); torch.cuda.synchronize()
print("accumulate_lanes 20: %.1f us"%(e0.elapsed_time(e1)/50*1e3))
PY
